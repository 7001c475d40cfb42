"""Device-side input stage (gca_clip_prepare + engine.input.DeviceInputStage) against oracle.input, which is pinned to the
reference's VideoNormalize / VideoToTensor by tests/golden/input.npz.  Byte / index work and a two-rounding affine map: the bar
is BIT-EXACT for fp32 output, one fp16 rounding of the exact fp32 value for fp16 output."""
import numpy as np
import pytest
import torch

import parity

pytestmark = pytest.mark.gpu
DEV = torch.device('cuda:0')


def _case(rng, b, views, T, Hs, Ws, H, W):
    frames = rng.randint(0, 256, size=(b, views, T, Hs, Ws, 3)).astype(np.uint8)
    params = np.zeros((b, views, 4), dtype=np.int32)
    params[..., 0] = rng.randint(0, Hs - H + 1, size=(b, views))
    params[..., 1] = rng.randint(0, Ws - W + 1, size=(b, views))
    params[..., 2] = rng.randint(0, 2, size=(b, views))
    return frames, params


@pytest.mark.parametrize('b,views,T,Hs,Ws,H,W', [
    (3, 2, 4, 20, 24, 16, 16),        # vector stores (W % 4 == 0), crops and flips
    (2, 2, 3, 13, 17, 9, 11),         # ragged W: scalar stores, tail columns
    (2, 1, 2, 8, 8, 8, 8),            # one view, identity crop
    (1, 2, 2, 5, 9, 1, 3),            # W < 4
    (4, 2, 8, 128, 171, 112, 112),    # the reference's frame geometry: 128 x 171 frames, 112 x 112 crops
])
def test_clip_prepare_bit_exact_vs_oracle(pkg, golden, b, views, T, Hs, Ws, H, W):
    from oracle import input as oinput
    inp = pkg.engine.input
    rng = np.random.RandomState(b * 100 + W)
    frames, params = _case(rng, b, views, T, Hs, Ws, H, W)
    frames[0, 0, 0, params[0, 0, 0], params[0, 0, 1]] = (0, 255, 128)
    for mean, std in (((0.485, 0.456, 0.406), (0.229, 0.224, 0.225)), ((0.5, 0.45, 0.4), (0.25, 0.3, 0.2))):
        want = oinput.make_batch(frames, params, H, W, mean, std)
        m, d = inp.normalize_constants(mean, std)
        mo, do = oinput.normalize_constants(mean, std)
        assert np.array_equal(m, mo) and np.array_equal(d, do)
        got = inp.clip_prepare(torch.from_numpy(frames).to(DEV), torch.from_numpy(params).to(DEV), m, d, H, W)
        assert got.dtype is torch.float32 and tuple(got.shape) == (b, 3 * views, T, H, W)
        assert torch.equal(got.cpu(), want)                                  # bit for bit
        got16 = inp.clip_prepare(torch.from_numpy(frames).to(DEV), torch.from_numpy(params).to(DEV), m, d, H, W,
                                 out_dtype=torch.float16)
        assert torch.equal(got16.cpu(), want.half())                         # the exact fp32 value, rounded once
    # the golden frames of the reference's own classes through the kernel (identity crop, no flip)
    g = golden('input')
    for tag in ('a', 'b'):
        fr = g.z[tag + ':frames']
        f6 = torch.from_numpy(fr[None, None]).contiguous().to(DEV)
        m, d = inp.normalize_constants(tuple(g.z[tag + ':mean']), tuple(g.z[tag + ':std']))
        out = inp.clip_prepare(f6, torch.zeros(1, 1, 4, dtype=torch.int32, device=DEV), m, d, fr.shape[1], fr.shape[2])
        assert torch.equal(out[0].cpu(), g.t(tag + ':tensor'))
    # properties that do not depend on size: a flip of a flipped source is the identity; a crop commutes with the stage
    f = torch.from_numpy(frames).to(DEV)
    p0 = torch.from_numpy(params).to(DEV)
    p1 = p0.clone(); p1[..., 2] ^= 1
    p1[..., 1] = (Ws - W) - p0[..., 1]
    a = inp.clip_prepare(f, p0, m, d, H, W)
    bb = inp.clip_prepare(f.flip(4).contiguous(), p1, m, d, H, W)
    assert torch.equal(a, bb)
    with pytest.raises(RuntimeError):
        inp.clip_prepare(torch.from_numpy(frames), p0, m, d, H, W)            # host frames: no CPU fallback
    with pytest.raises(ValueError):
        inp.clip_prepare(f.float(), p0, m, d, H, W)


@pytest.mark.parametrize('kind', ['moco', 'simsiam'])
def test_trainer_consumes_staged_uint8_batches(pkg, kind):
    """train_step(StagedBatch): pinned double-buffered H2D on the copy stream + the prepare kernel writing the trainer's static
    batch == train_step on the oracle-built fp32 batch, bit for bit (same kernels downstream), over enough steps to wrap the
    two slots and to run eager, capture and replay."""
    from oracle import input as oinput
    parity.register_tiny(pkg)
    b, T, Hs, Ws, S = 8, 8, 56, 60, 48
    rng = np.random.RandomState(3)
    batches = [_case(rng, b, 2, T, Hs, Ws, S, S) for _ in range(6)]
    shs = [torch.randperm(b, generator=torch.Generator().manual_seed(i)) for i in range(6)]

    def make():
        if kind == 'moco':
            return pkg.MoCoTrainer(parity.make_cfg(pkg, 'R2P1D10T', 'moco', 32, 64, T), DEV, use_graph=True, seed=4)
        return pkg.SimSiamTrainer(parity.make_cfg(pkg, 'R2P1D10T', 'simsiam', 32, 16, T), DEV, use_graph=True, seed=4)
    step = (lambda tr, x, i: tr.train_step(x, shuffle_ids=shs[i])) if kind == 'moco' else (lambda tr, x, i: tr.train_step(x))
    ref, tr = make(), make()
    stage = pkg.engine.input.DeviceInputStage(b, T, (Hs, Ws), S, DEV)
    nxt = stage.stage(*batches[0])
    for i, (frames, params) in enumerate(batches):
        cur = nxt
        if i + 1 < len(batches):
            nxt = stage.stage(*batches[i + 1])                   # batch i+1 is copied while step i runs
        o1 = step(tr, cur, i)
        o2 = step(ref, oinput.make_batch(frames, params, S, S).to(DEV), i)
        assert torch.equal(o1['loss'], o2['loss']), i
        if kind == 'moco':
            assert torch.equal(o1['logits'], o2['logits']) and torch.equal(o1['q'], o2['q']), i
    a1 = tr.arena_q if kind == 'moco' else tr.arena
    a2 = ref.arena_q if kind == 'moco' else ref.arena
    assert torch.equal(a1.flat, a2.flat)
    assert tr._segments[0].graph is not None
    with pytest.raises(ValueError):
        bad = batches[0][1].copy(); bad[0, 0, 0] = Hs            # crop window outside the frame
        stage.stage(batches[0][0], bad)
    tr.close(); ref.close()
