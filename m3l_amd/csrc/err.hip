// Thread-local error string of the C ABI (include/m3l_amd.h: m3l_last_error); every launcher reports through m3l_set_error.
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include "../../include/m3l_amd.h"
#include "common.cuh"

static thread_local char g_err[512] = {0};
void m3l_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" int m3l_last_error(char* buf, size_t n) {
    if (buf && n) {
        strncpy(buf, g_err, n - 1);
        buf[n - 1] = 0;
    }
    return (int)strlen(g_err);
}

// residual-stream mode of the launch being issued (common.cuh): the transformer plan sets it around the kernels of a stack whose every
// layer runs on kernels that support a bf16 residual stream; thread-local, as forward and backward run on different threads under autograd
static thread_local int g_call_rb = 0;
int m3l_call_rb(void) { return g_call_rb; }
void m3l_set_call_rb(int rb) { g_call_rb = rb ? 1 : 0; }

// the tensors a stack exchanges with its neighbours in the fused step (decoder input written by the un-shuffle, its gradient read by the
// un-shuffle backward) are bf16 as well: set by mae_step.hip around the calls concerned when that stack runs the bf16 residual stream
static thread_local int g_call_io = 0;
int m3l_call_io(void) { return g_call_io; }
void m3l_set_call_io(int on) { g_call_io = on ? 1 : 0; }
