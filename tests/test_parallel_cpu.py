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
    # finish() is idempotent: a second call, or a call with nothing sent since the last one, must not scale again
    sync.zero_grad()
    sync.flat.fill_(float(rank + 1))
    sync.finish()
    sync.finish()
    ref = torch.full_like(sync.flat, float(rank + 1))
    dist.all_reduce(ref)
    ref.mul_(1.0 / world)
    assert torch.equal(sync.flat, ref), (float(sync.flat[0]), float(ref[0]))
    # deferred scale: the sums stay in the buffer, 1 / world is handed over exactly once (FlatAdam folds it into its launch)
    sync.zero_grad()
    sync.flat.fill_(float(rank + 1))
    sync.finish(defer_scale=True)
    assert torch.equal(sync.flat, torch.full_like(sync.flat, 3.0))
    assert sync.take_scale() == 1.0 / world and sync.take_scale() == 1.0
    sync.zero_grad()
    sync.flat.fill_(float(rank + 1))
    sync.finish(defer_scale=True)
    sync.finish()                                    # nobody consumed the deferred factor: a plain finish() applies it
    assert torch.equal(sync.flat, torch.full_like(sync.flat, 1.5)) and sync.take_scale() == 1.0
    ret[rank] = float(sync.flat.sum())
    # learned position tables (use_sincosmod_encodings=False, pretrain_models.py:218-219,280-287): trained, trailing span of the buffer
    torch.manual_seed(0)
    enc2 = m3l_amd.VTT(image_size=32, tactile_size=16, image_patch_size=8, tactile_patch_size=4, dim=64, depth=1, heads=2, mlp_dim=128)
    mae2 = m3l_amd.VTMAE(encoder=enc2, decoder_dim=64, decoder_depth=1, decoder_heads=2, use_sincosmod_encodings=False)
    sync2 = GradSync(mae2)
    n2 = dict(mae2.named_parameters())
    pe, dpe = n2["encoder.pos_embedding"], n2["decoder_pos_emb.weight"]
    assert pe.grad is not None and dpe.grad is not None
    tail = sorted([sync2._span[id(pe)], sync2._span[id(dpe)]])
    assert tail[0][1] == tail[1][0] and tail[1][1] == sync2.flat.numel(), tail          # the trailing span
    assert sync2.buckets[-1][0] == tail[0][0] and 5 in sync2._bucket_ids
    sync2.zero_grad()
    for b in sync2._bucket_ids[:-1]:                 # the modules report their buckets; the position tables arrive through autograd
        s_, e_, _ = sync2.buckets[sync2._bucket_ids.index(b)]
        sync2.flat[s_:e_] = float(rank + 1)
        sync2.bucket_done(b)
    pe.grad.add_(float(rank + 1))                    # what AccumulateGrad does with the batch-summed token gradients
    dpe.grad.add_(float(rank + 1))
    sync2.finish()
    assert torch.allclose(sync2.flat, torch.full_like(sync2.flat, 1.5))
    assert float(pe.grad.flatten()[0]) == 1.5 and float(dpe.grad[0, 0]) == 1.5
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
