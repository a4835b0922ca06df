"""Multi-process (world_size 2, gloo, CPU) test of the gradient exchange used by the N > 1 path: flat bucketed buffer,
per-bucket asynchronous all-reduce fired in backward order, SUM + 1/world scaling, unused parameters skipped."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import m3l_amd
from m3l_amd.parallel import GradSync, _bucket_of


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    enc = m3l_amd.VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=1, heads=2, mlp_dim=128)
    mae = m3l_amd.VTMAE(encoder=enc, decoder_dim=64, decoder_depth=1, decoder_heads=2)
    sync = GradSync(mae)
    assert sync.world == world
    # every trainable parameter has a grad view in the flat buffer; the two sincos-unused ones do not
    names = dict(mae.named_parameters())
    assert names["encoder.pos_embedding"].grad is None and names["decoder_pos_emb.weight"].grad is None
    assert sum(p.numel() for p in sync.params) == sync.flat.numel()
    assert all(p.grad.data_ptr() >= sync.flat.data_ptr() for p in sync.params)
    # buckets follow the backward order heads -> decoder -> glue -> encoder -> embed and tile the buffer
    assert [b[0] for b in sync.buckets][0] == 0 and sync.buckets[-1][1] == sync.flat.numel()
    assert all(sync.buckets[i][1] == sync.buckets[i + 1][0] for i in range(len(sync.buckets) - 1))
    # emulate the backward: each module writes its slice, then reports its bucket
    sync.zero_grad()
    for b in (0, 1, 2, 3, 4):
        s, e, _ = sync.buckets[sync._bucket_ids.index(b)]
        sync.flat[s:e] = float(rank + 1) * (b + 1)
        sync.bucket_done(b)
    sync.finish()
    for b in (0, 1, 2, 3, 4):
        s, e, _ = sync.buckets[sync._bucket_ids.index(b)]
        expect = (b + 1) * sum(r + 1 for r in range(world)) / world
        assert torch.allclose(sync.flat[s:e], torch.full((e - s,), expect)), (b, float(sync.flat[s]), expect)
    # chunked transformer bucket: layers are laid out top-down, a chunk is one contiguous slice, only that slice is reduced
    sync.zero_grad()
    enc_params = [p for n_, p in mae.named_parameters() if n_.startswith("encoder.transformer")]
    spans = sorted(sync._span[id(p)] for p in enc_params)
    assert all(spans[i][1] == spans[i + 1][0] for i in range(len(spans) - 1))
    norm_w = dict(mae.named_parameters())["encoder.transformer.norm.weight"]
    assert sync._span[id(norm_w)][0] == spans[0][0]                       # final norm first (its backward runs first)
    sync.flat.fill_(float(rank + 1))
    sync.range_done(3, enc_params, last=True)
    sync.finish()       # also reduces the buckets nobody reported
    assert torch.allclose(sync.flat, torch.full_like(sync.flat, 1.5))
    # parameters see the reduced values through their .grad views
    assert float(mae.to_pixels.weight.grad[0, 0]) == 1.5
    ret[rank] = float(sync.flat.sum())
    dist.destroy_process_group()


def test_gradsync_two_ranks_gloo():
    world = 2
    port = _free_port()
    with mp.Manager() as m:
        ret = m.dict()
        mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
        assert len(ret) == world and abs(ret[0] - ret[1]) < 1e-3


def test_bucket_assignment():
    assert _bucket_of("to_pixels.weight") == 0 and _bucket_of("to_tactiles.bias") == 0
    assert _bucket_of("decoder.layers.0.0.to_qkv.weight") == 1 and _bucket_of("decoder.norm.weight") == 1
    assert _bucket_of("decoder_modality_embedding.weight") == 2 and _bucket_of("mask_token") == 2 and _bucket_of("enc_to_dec.weight") == 2
    assert _bucket_of("encoder.transformer.layers.3.1.net.1.weight") == 3
    assert _bucket_of("encoder.image_to_patch_embedding.2.weight") == 4 and _bucket_of("encoder_modality_embedding.weight") == 4
