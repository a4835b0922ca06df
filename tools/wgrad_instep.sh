#!/bin/bash
# in-step and stand-alone duration of the grouped weight-gradient launch under environment settings: bash tools/wgrad_instep.sh "VAR=v ..." ...
for setting in "$@"; do
  echo -n "[$setting] "
  env $setting python3 tools/prof_stats_lite.py "wgrad["
done
