"""GPU (-m gpu): every HIP kernel of libdlsg_hip.so against the torch emulation of the same interface, on identical
views (strides, offsets, ragged edges).  Tolerances are fp32: 2e-5 relative to the output scale unless stated."""
import math

import numpy as np
import pytest
import torch

from emul_ops import EmulOps, GEMM_NT, GEMM_NN, GEMM_TN, F_ACCUM, F_TANH, F_BF16X3, F_FORCE64, F_FORCE128, F_TILE256

F_SK, F_NOSK = 4096, 8192          # include/dlsg.h DLSG_GEMM_SK / DLSG_GEMM_NOSK (dlsg_amd.hip needs the library: not imported here)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def hip():
    from dlsg_amd.hip import HipOps
    return HipOps()


def rnd(gen, *shape, scale=1.0):
    return torch.randn(*shape, generator=gen) * scale


def both(hip, build, run, outs, tol=2e-5, name=''):
    """build(gen) -> dict of CPU base tensors; run(ops, t) executes the op on dict t (views are made inside run);
    outs = names of tensors to compare."""
    gen = torch.Generator().manual_seed(1234)
    base = build(gen)
    tc = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in base.items()}
    tg = {k: (v.cuda() if torch.is_tensor(v) else v) for k, v in base.items()}
    run(EmulOps(), tc)
    run(hip, tg)
    torch.cuda.synchronize()
    for o in outs:
        a, b = tc[o], tg[o].cpu()
        if a.dtype in (torch.int64, torch.int32):
            assert torch.equal(a, b), (name, o)
            continue
        assert torch.isfinite(b).all(), (name, o, 'non-finite')
        err = (a - b).abs().max().item()
        ref = max(a.abs().max().item(), 1e-6)
        assert err <= tol * max(ref, 1.0) + tol, (name, o, 'max err %g (ref scale %g)' % (err, ref))


GEMM_SHAPES = [
    # M, N, K
    (64, 64, 32), (128, 128, 64), (1, 1, 1), (3, 50, 244), (70, 33, 100), (200, 130, 37), (257, 129, 65),
    (26, 26, 2048), (416, 26, 64), (64, 4096, 300), (512, 1000, 96),
]


@pytest.mark.parametrize('mode', [GEMM_NT, GEMM_NN, GEMM_TN])
@pytest.mark.parametrize('shape', GEMM_SHAPES)
def test_gemm_plain(hip, mode, shape):
    M, N, K = shape

    def build(g):
        if mode == GEMM_NT:
            A, B = rnd(g, M, K), rnd(g, N, K)
        elif mode == GEMM_NN:
            A, B = rnd(g, M, K), rnd(g, K, N)
        else:
            A, B = rnd(g, K, M), rnd(g, K, N)
        return dict(A=A, B=B, C=torch.zeros(M, N), bias=rnd(g, N))

    def run(ops, t):
        ops.gemm(mode, [(t['A'], t['B'], t['C'])], alpha=0.5, bias=t['bias'])
    both(hip, build, run, ['C'], tol=1e-5 * max(1.0, math.sqrt(K)), name='gemm%d %s' % (mode, shape))


@pytest.mark.parametrize('mode', [GEMM_NT, GEMM_NN, GEMM_TN])
@pytest.mark.parametrize('shape', [(128, 64, 64), (3, 50, 244), (257, 129, 65), (300, 1000, 96), (1664, 200, 130)])
def test_gemm_128x64_tile(hip, mode, shape):
    """the mid-size tile (both FORCE bits): ragged edges, unaligned strides, bias + accumulate, two groups of different width"""
    M, N, K = shape

    def build(g):
        if mode == GEMM_NT:
            A, B = rnd(g, M, K + 3), rnd(g, N, K + 3)
        elif mode == GEMM_NN:
            A, B = rnd(g, M, K + 3), rnd(g, K, N + 3)
        else:
            A, B = rnd(g, K, M + 3), rnd(g, K, N + 3)
        return dict(A=A, B=B, C=rnd(g, M, N + 5), C2=rnd(g, M, N + 5), bias=rnd(g, N))

    def run(ops, t):
        N2 = max(1, N // 2)
        if mode == GEMM_NT:
            A, B, B2 = t['A'][:, :K], t['B'][:, :K], t['B'][:N2, :K]
        elif mode == GEMM_NN:
            A, B, B2 = t['A'][:, :K], t['B'][:, :N], t['B'][:, :N2]
        else:
            A, B, B2 = t['A'][:, :M], t['B'][:, :N], t['B'][:, :N2]
        ops.gemm(mode, [(A, B, t['C'][:, :N], t['bias']), (A, B2, t['C2'][:, :N2], t['bias'][:N2])], alpha=0.5,
                 flags=F_ACCUM | F_FORCE64 | F_FORCE128)
    both(hip, build, run, ['C', 'C2'], tol=1e-5 * max(1.0, math.sqrt(K)), name='gemm 128x64 %d %s' % (mode, shape))


@pytest.mark.parametrize('mode', [GEMM_NT, GEMM_NN, GEMM_TN])
@pytest.mark.parametrize('tile', [F_TILE256, F_TILE256 | F_FORCE128])
@pytest.mark.parametrize('shape', [(256, 256, 32), (600, 520, 100), (260, 132, 36), (512, 1000, 96)])
def test_gemm_256_tiles(hip, mode, tile, shape):
    """csrc/gemm_big.hip (256 x 256 and 256 x 128 tiles, LDS-DMA stages): ragged edges, a partial last K stage, strided views,
    bias + accumulate, two groups of different width"""
    M, N, K = shape

    def build(g):
        if mode == GEMM_NT:
            A, B = rnd(g, M, K + 4), rnd(g, N, K + 4)
        elif mode == GEMM_NN:
            A, B = rnd(g, M, K + 4), rnd(g, K, N + 4)
        else:
            A, B = rnd(g, K, M + 4), rnd(g, K, N + 4)
        return dict(A=A, B=B, C=rnd(g, M, N + 5), C2=rnd(g, M, N + 5), bias=rnd(g, N))

    def run(ops, t):
        N2 = max(4, N // 2 // 4 * 4)
        if mode == GEMM_NT:
            A, B, B2 = t['A'][:, :K], t['B'][:, :K], t['B'][:N2, :K]
        elif mode == GEMM_NN:
            A, B, B2 = t['A'][:, :K], t['B'][:, :N], t['B'][:, :N2]
        else:
            A, B, B2 = t['A'][:, :M], t['B'][:, :N], t['B'][:, :N2]
        ops.gemm(mode, [(A, B, t['C'][:, :N], t['bias']), (A, B2, t['C2'][:, :N2], t['bias'][:N2])], alpha=0.5, flags=F_ACCUM | tile)
    both(hip, build, run, ['C', 'C2'], tol=1e-5 * max(1.0, math.sqrt(K)), name='gemm 256 tile %d %s' % (mode, shape))


@pytest.mark.parametrize('mode', [GEMM_NT, GEMM_NN, GEMM_TN])
@pytest.mark.parametrize('extra', [0, F_ACCUM, F_TANH])
@pytest.mark.parametrize('bm', [0, 16384, 32768, 262144])    # the dispatcher's tile, DLSG_GEMM_SK_BM128, _BM256, _BN128 (128 x 128)
@pytest.mark.parametrize('shape', [(256, 256, 32), (600, 520, 96), (260, 132, 64), (1700, 1000, 128), (300, 260, 2048)])
def test_gemm_stream_k(hip, mode, extra, bm, shape):
    """csrc/gemm_sk.hip (persistent stream-K launch, forced with F_SK): ragged edges, strided views, per-group bias, two groups of
    different width, store / accumulate / tanh epilogues; tiles cut between 2 .. 14 workgroups (few tiles on 256 CUs), whole
    tiles plus a cut remainder; twice in a row (the counters in the workspace must come back to zero)"""
    M, N, K = shape

    def build(g):
        if mode == GEMM_NT:
            A, B = rnd(g, M, K + 4), rnd(g, N, K + 4)
        elif mode == GEMM_NN:
            A, B = rnd(g, M, K + 4), rnd(g, K, N + 4)
        else:
            A, B = rnd(g, K, M + 4), rnd(g, K, N + 4)
        return dict(A=A, B=B, C=rnd(g, M, N + 8), C2=rnd(g, M, N + 8), bias=rnd(g, N))

    def run(ops, t):
        N2 = max(4, N // 2 // 4 * 4)
        if mode == GEMM_NT:
            A, B, B2 = t['A'][:, :K], t['B'][:, :K], t['B'][:N2, :K]
        elif mode == GEMM_NN:
            A, B, B2 = t['A'][:, :K], t['B'][:, :N], t['B'][:, :N2]
        else:
            A, B, B2 = t['A'][:, :M], t['B'][:, :N], t['B'][:, :N2]
        groups = [(A, B, t['C'][:, :N], t['bias']), (A, B2, t['C2'][:, :N2], t['bias'][:N2])]
        ops.gemm(mode, groups, alpha=0.5, flags=extra | F_SK | bm)
        if not (extra & F_ACCUM):
            ops.gemm(mode, groups, alpha=0.5, flags=extra | F_SK | bm)
    both(hip, build, run, ['C', 'C2'], tol=(1e-4 if extra & F_TANH else 1e-5) * max(1.0, math.sqrt(K)), name='gemm stream-K %d %s' % (mode, shape))
    ws = hip._gemm_workspace(torch.device('cuda', 0))
    assert int(ws[:1024].view(torch.int32).abs().sum().item()) == 0          # the counter area
    assert int(hip._persist_word(torch.device('cuda', 0)).item()) == 0


def test_gemm_stream_k_is_bit_reproducible_and_refuses_what_it_cannot_take(hip):
    A, B = torch.randn(3000, 2048, device='cuda'), torch.randn(768, 2048, device='cuda')
    C1, C2 = torch.empty(3000, 768, device='cuda'), torch.empty(3000, 768, device='cuda')
    hip.gemm(GEMM_NT, [(A, B, C1)], flags=F_SK)
    hip.gemm(GEMM_NT, [(A, B, C2)], flags=F_SK)
    assert torch.equal(C1, C2)
    ref = (A.double() @ B.double().t())
    assert ((C1.double() - ref).abs().max() / ref.abs().max()).item() < 2e-6
    with pytest.raises(RuntimeError):
        hip.gemm(GEMM_NT, [(A[:, :2040], B[:, :2040], C1)], flags=F_SK)      # K % 32 != 0
    with pytest.raises(RuntimeError):
        hip.gemm(GEMM_NT, [(A, B[:766], C1[:, :766])], flags=F_SK)           # width not a multiple of 4


@pytest.mark.parametrize('mode,extra,shape', [(GEMM_NT, F_TANH, (600, 520, 512)), (GEMM_TN, F_ACCUM, (300, 260, 2048)),
                                              (GEMM_NN, 0, (1700, 1000, 128)), (GEMM_TN, F_ACCUM, (1024, 2048, 6656))])
def test_gemm_stream_k_without_waiting(hip, mode, extra, shape):
    """No workgroup of the stream-K launch waits for another beyond a bounded poll: a contributor that finds its split tile
    incomplete gives its sub-blocks to the one that decides last.  DLSG_GEMM_SK_GIVEAWAY forces that path for every
    contributor; the result is the same bit for bit (same contributor-order sums) and the counters are back at zero."""
    M, N, K = shape
    g = torch.Generator(device='cuda').manual_seed(3)
    if mode == GEMM_NT:
        A, B = torch.randn(M, K, device='cuda', generator=g), torch.randn(N, K, device='cuda', generator=g)
    elif mode == GEMM_NN:
        A, B = torch.randn(M, K, device='cuda', generator=g), torch.randn(K, N, device='cuda', generator=g)
    else:
        A, B = torch.randn(K, M, device='cuda', generator=g), torch.randn(K, N, device='cuda', generator=g)
    C0 = torch.randn(M, N, device='cuda', generator=g)
    bias = torch.randn(N, device='cuda', generator=g)
    out = []
    for give in (0, 65536, 0, 65536):
        for bm in (16384, 32768, 262144):
            Cc = C0.clone()
            hip.gemm(mode, [(A, B, Cc, bias)], flags=extra | F_SK | bm | give)
            out.append(Cc)
    torch.cuda.synchronize()
    for i in (0, 1, 2):
        assert torch.equal(out[i], out[i + 3]) and torch.equal(out[i], out[i + 6]) and torch.equal(out[i], out[i + 9])
    ws = hip._gemm_workspace(torch.device('cuda', 0))
    assert int(ws[:1024].view(torch.int32).abs().sum().item()) == 0


@pytest.mark.parametrize('mode,M,N,K,G', [(GEMM_TN, 1024, 2048, 1024, 2), (GEMM_NT, 2048, 4096, 512, 1), (GEMM_NN, 1024, 2048, 768, 1)])
def test_gemm_stream_k_slice_per_xcd_map(hip, mode, M, N, K, G):
    """Launches without whole-tile rounds whose tiles all have 4, 2 or 8 contributors run with an XCD on ONE K-slice of a block of
    tiles (csrc/gemm_sk.hip, SkArgs::xmap); DLSG_GEMM_SK_NOXMAP keeps every tile's contributors on one XCD.  The map only permutes
    which workgroup does what: same bits either way, and both agree with fp64."""
    F_NOXMAP = 131072
    g = torch.Generator(device='cuda').manual_seed(7)
    groups, refs = [], []
    for _ in range(G):
        if mode == GEMM_NT:
            A, B = torch.randn(M, K, device='cuda', generator=g), torch.randn(N, K, device='cuda', generator=g)
            r = A.double() @ B.double().t()
        elif mode == GEMM_NN:
            A, B = torch.randn(M, K, device='cuda', generator=g), torch.randn(K, N, device='cuda', generator=g)
            r = A.double() @ B.double()
        else:
            A, B = torch.randn(K, M, device='cuda', generator=g), torch.randn(K, N, device='cuda', generator=g)
            r = A.double().t() @ B.double()
        groups.append((A, B))
        refs.append(r)
    outs = []
    for fl in (0, F_NOXMAP):
        Cs = [torch.full((M, N), float('nan'), device='cuda') for _ in range(G)]
        hip.gemm(mode, [(a_, b_, c_) for (a_, b_), c_ in zip(groups, Cs)], flags=F_SK | 32768 | fl)      # 256 x 256 tiles
        outs.append(Cs)
    torch.cuda.synchronize()
    for c0, c1, r in zip(outs[0], outs[1], refs):
        assert torch.equal(c0, c1)
        assert ((c0.double() - r).abs().max() / r.abs().max()).item() < 3e-6
    ws = hip._gemm_workspace(torch.device('cuda', 0))
    assert int(ws[:1024].view(torch.int32).abs().sum().item()) == 0


@pytest.mark.parametrize('mode,M,N,K,G', [(GEMM_TN, 1024, 2048, 6656, 2), (GEMM_NT, 3000, 1024, 2048, 1), (GEMM_NN, 1664, 2048, 2048, 3)])
@pytest.mark.parametrize('budget', [224, 100, 8, 3, 100000])
def test_gemm_stream_k_on_a_cu_budget(hip, mode, M, N, K, G, budget):
    """dlsg_gemm_args.cu_budget (ABI 8): a stream-K launch with at most that many workgroups (rounded down to a multiple of 8, at
    least 8; more than the chip has = every CU) -- the CUs it leaves are for a co-tenant, a collective's kernels on another
    stream.  Any budget gives the fp64 answer, the same bits twice in a row, and leaves the workspace counters at zero."""
    g = torch.Generator(device='cuda').manual_seed(11)
    groups, refs = [], []
    for _ in range(G):
        if mode == GEMM_NT:
            A, B = torch.randn(M, K, device='cuda', generator=g), torch.randn(N, K, device='cuda', generator=g)
            r = A.double() @ B.double().t()
        elif mode == GEMM_NN:
            A, B = torch.randn(M, K, device='cuda', generator=g), torch.randn(K, N, device='cuda', generator=g)
            r = A.double() @ B.double()
        else:
            A, B = torch.randn(K, M, device='cuda', generator=g), torch.randn(K, N, device='cuda', generator=g)
            r = A.double().t() @ B.double()
        groups.append((A, B))
        refs.append(r)
    outs = []
    hip.sk_cu_budget = budget
    try:
        for _ in range(2):
            Cs = [torch.full((M, N), float('nan'), device='cuda') for _ in range(G)]
            hip.gemm(mode, [(a_, b_, c_) for (a_, b_), c_ in zip(groups, Cs)], flags=F_SK)
            outs.append(Cs)
    finally:
        hip.sk_cu_budget = 0
    torch.cuda.synchronize()
    for c0, c1, r in zip(outs[0], outs[1], refs):
        assert torch.equal(c0, c1)
        # fp32 chains of K terms: the error grows like sqrt(K) (3.4e-6 at K = 6656, whatever the budget)
        assert ((c0.double() - r).abs().max() / r.abs().max()).item() < 2e-6 + 6e-8 * math.sqrt(K)
    ws = hip._gemm_workspace(torch.device('cuda', 0))
    assert int(ws[:1024].view(torch.int32).abs().sum().item()) == 0


def test_gemm_variant_names_the_tile_family(hip):
    """dlsg_gemm_variant == the choice dlsg_gemm makes (include/dlsg.h DLSG_GEMM_V_*), on the shapes DESIGN.md quotes"""
    def plan(mode, M, N, K, G=1, flags=0):
        dev = 'cuda'
        if mode == GEMM_NT:
            A, B = torch.empty(M, K, device=dev), torch.empty(N, K, device=dev)
        elif mode == GEMM_NN:
            A, B = torch.empty(M, K, device=dev), torch.empty(K, N, device=dev)
        else:
            A, B = torch.empty(K, M, device=dev), torch.empty(K, N, device=dev)
        Cc = torch.empty(M, N, device=dev)
        return hip.gemm(mode, [(A, B, Cc)] * G, flags=flags, plan_only=True)
    # with the caller's workspace the chip-filling products are one persistent stream-K launch (csrc/gemm_sk.hip) ...
    assert plan(GEMM_NT, 26624, 1024, 2048, 2) == 7              # region projections of both streams
    assert plan(GEMM_TN, 1024, 2048, 26624, 2) == 7              # their weight gradients: the contraction is cut inside the launch
    assert plan(GEMM_TN, 4096, 1024, 1664, 11) == 7              # the 4096-row weight-gradient blocks of the decoder and the BiLSTM
    assert plan(GEMM_NT, 1664, 2048, 2048, 3) == 7               # the 1 664-row products of the step (on 128-row tiles) ...
    assert plan(GEMM_NT, 1664, 1024, 2048) in (0, 1)             # ... down to ~80 us per workgroup: below that the small tiles
    assert plan(GEMM_NT, 1664, 1000, 1024) != 7                  # (the vocabulary projection)
    assert plan(GEMM_NT, 26624, 1024, 2046, 2) == 2              # K % 32 != 0: not a stream-K shape
    assert plan(GEMM_NT, 832, 1024, 2048, 2) != 7                # 832 rows (the goldens' two clips): too little work, 19 % padding
    # groups of different depth whose tiles do not fill whole rounds are not dealt out evenly: left to the tiled kernels
    x, w1, w2 = torch.empty(1664, 6144, device='cuda'), torch.empty(1024, 6144, device='cuda'), torch.empty(1024, 2048, device='cuda')
    c1, c2 = torch.empty(1664, 1024, device='cuda'), torch.empty(1664, 1024, device='cuda')
    assert hip.gemm(GEMM_NT, [(x, w1, c1), (x[:, :2048], w2, c2)], plan_only=True) != 7
    assert hip.gemm(GEMM_NT, [(x, w1, c1), (x, w1, c2)], plan_only=True) == 7
    # ... without it (F_NOSK: the binding passes no workspace) the tiled kernels' rule is what it was
    _plan = plan
    plan = lambda mode, M, N, K, G=1, flags=0: _plan(mode, M, N, K, G, flags | F_NOSK)
    assert plan(GEMM_NT, 64, 4096, 1024, 4) == 3                 # recurrent product: skinny kernels
    assert plan(GEMM_NT, 128, 4096, 1024, 4) == 3
    assert plan(GEMM_TN, 64, 4096, 1024) != 3                    # (row-contiguous A: not a skinny shape)
    assert plan(GEMM_NT, 200, 130, 37) == 0                      # 64 x 64
    assert plan(GEMM_NT, 1664, 2048, 2048, 3) == 1               # 128 x 64
    assert plan(GEMM_TN, 2048, 2048, 1664, 3) == 2               # 128 x 128
    assert plan(GEMM_TN, 1024, 2048, 3328, 16) == 4              # whole launch on 256 x 256: 512 tiles = two rounds of the CUs
    assert plan(GEMM_NT, 26624, 1024, 2048, 2) == 6              # 96 row panels in three rounds on 256 x 256, 8 panels on smaller tiles
    assert plan(GEMM_NT, 26624, 1024, 2046, 2) == 2              # K % 4 != 0: the LDS-DMA tiles do not take it
    assert plan(GEMM_NT, 512, 512, 64, flags=F_TILE256) == 4 and plan(GEMM_NT, 512, 512, 64, flags=F_TILE256 | F_FORCE128) == 5


def test_gemm_256_tiles_refuse_unaligned_operands(hip):
    A, B, C = torch.randn(256, 30, device='cuda'), torch.randn(256, 30, device='cuda'), torch.zeros(256, 256, device='cuda')
    with pytest.raises(RuntimeError):
        hip.gemm(GEMM_NT, [(A, B, C)], flags=F_TILE256)          # K = 30: not a multiple of 4, shorter than a stage


@pytest.mark.parametrize('mode', [GEMM_NT, GEMM_NN, GEMM_TN])
def test_gemm_whole_rounds_on_256_tiles_and_the_rest(hip, mode):
    """a launch of >= 1000 tiles whose row panels do not come in whole rounds of the 256 CUs: the dispatcher multiplies the
    first 8192 rows on 256 x 256 tiles and the remaining 300 through the smaller tiles (csrc/gemm.hip, dlsg_gemm)"""
    M, N, K, G = 8492, 1024, 64, 2

    def build(g):
        d = {}
        for i in range(G):
            if mode == GEMM_NT:
                d['A%d' % i], d['B%d' % i] = rnd(g, M, K), rnd(g, N, K)
            elif mode == GEMM_NN:
                d['A%d' % i], d['B%d' % i] = rnd(g, M, K), rnd(g, K, N)
            else:
                d['A%d' % i], d['B%d' % i] = rnd(g, K, M), rnd(g, K, N)
            d['C%d' % i] = rnd(g, M, N)
        d['bias'] = rnd(g, N)
        return d

    def run(ops, t):
        ops.gemm(mode, [(t['A%d' % i], t['B%d' % i], t['C%d' % i]) for i in range(G)], bias=t['bias'], flags=F_ACCUM | F_TANH)
    both(hip, build, run, ['C%d' % i for i in range(G)], tol=2e-5, name='head/tail gemm %d' % mode)


@pytest.mark.parametrize('mode', [GEMM_NT, GEMM_NN, GEMM_TN])
def test_gemm_many_groups_of_different_width(hip, mode):
    """a grouped launch of >= 1000 tiles whose groups differ in width (the decoder's weight-gradient blocks), with C += and alpha"""
    M, N, K = 2048, 1024, 64
    widths = [1024, 1024, 300, 1024, 1024, 1024, 1024, 1024, 1024, 1024]

    def build(g):
        d = {}
        for i, wd in enumerate(widths):
            if mode == GEMM_NT:
                d['A%d' % i], d['B%d' % i] = rnd(g, M, K), rnd(g, wd, K)
            elif mode == GEMM_NN:
                d['A%d' % i], d['B%d' % i] = rnd(g, M, K), rnd(g, K, wd + (4 - wd % 4) % 4)
            else:
                d['A%d' % i], d['B%d' % i] = rnd(g, K, M), rnd(g, K, wd + (4 - wd % 4) % 4)
            d['C%d' % i] = rnd(g, M, wd)
        return d

    def run(ops, t):
        groups = []
        for i, wd in enumerate(widths):
            B = t['B%d' % i] if mode == GEMM_NT else t['B%d' % i][:, :wd]
            groups.append((t['A%d' % i], B, t['C%d' % i]))
        ops.gemm(mode, groups, alpha=0.5, flags=F_ACCUM)
    both(hip, build, run, ['C%d' % i for i in range(len(widths))], tol=2e-5, name='group head/tail gemm %d' % mode)


@pytest.mark.parametrize('mode', [GEMM_NT, GEMM_NN, GEMM_TN])
@pytest.mark.parametrize('shape', [(1664, 2048, 40, 3), (1000, 2048, 64, 2), (1664, 2048, 24, 4)])
def test_gemm_mid_size_tile_choice(hip, mode, shape):
    """launches whose tile the dispatcher picks itself: 624 / 256 (128x64) and 832 (128x128) 128-square tiles"""
    M, N, K, G = shape

    def build(g):
        d = {}
        for i in range(G):
            if mode == GEMM_NT:
                d['A%d' % i], d['B%d' % i] = rnd(g, M, K), rnd(g, N, K)
            elif mode == GEMM_NN:
                d['A%d' % i], d['B%d' % i] = rnd(g, M, K), rnd(g, K, N)
            else:
                d['A%d' % i], d['B%d' % i] = rnd(g, K, M), rnd(g, K, N)
            d['C%d' % i] = torch.zeros(M, N)
        return d

    def run(ops, t):
        ops.gemm(mode, [(t['A%d' % i], t['B%d' % i], t['C%d' % i]) for i in range(G)])
    both(hip, build, run, ['C%d' % i for i in range(G)], tol=1e-5 * max(1.0, math.sqrt(K)), name='mid gemm %d %s' % (mode, shape))


@pytest.mark.parametrize('mode', [GEMM_NT, GEMM_NN, GEMM_TN])
@pytest.mark.parametrize('force', [F_FORCE64, F_FORCE128])
@pytest.mark.parametrize('shape', [(128, 128, 64), (3, 50, 244), (70, 33, 100), (257, 129, 65), (416, 26, 64), (300, 1000, 96)])
def test_gemm_bf16x3(hip, mode, force, shape):
    """split-bf16 path: three bf16 MFMAs per product; error budget ~1e-5 per product (see csrc/gemm_bf16x3.hip)."""
    M, N, K = shape

    def build(g):
        if mode == GEMM_NT:
            A, B = rnd(g, M, K + 4), rnd(g, N, K + 4)
        elif mode == GEMM_NN:
            A, B = rnd(g, M, K + 4), rnd(g, K, N + 4)
        else:
            A, B = rnd(g, K, M + 4), rnd(g, K, N + 4)
        return dict(A=A, B=B, C=rnd(g, M, N + 4), bias=rnd(g, N))

    def run(ops, t):
        if mode == GEMM_NT:
            A, B = t['A'][:, :K], t['B'][:, :K]
        elif mode == GEMM_NN:
            A, B = t['A'][:, :K], t['B'][:, :N]
        else:
            A, B = t['A'][:, :M], t['B'][:, :N]
        ops.gemm(mode, [(A, B, t['C'][:, :N])], alpha=0.5, bias=t['bias'], flags=F_BF16X3 | F_ACCUM | force)
    both(hip, build, run, ['C'], tol=3e-5 * max(1.0, math.sqrt(K)), name='x3 gemm%d %s' % (mode, shape))


def test_gemm_bf16x3_error_is_compensated(hip):
    """The three-term split must be ~100x more accurate than a plain bf16 product (which would be ~4e-3)."""
    g = torch.Generator().manual_seed(0)
    A, B = rnd(g, 512, 1024), rnd(g, 512, 1024)
    ref = (A.double() @ B.double().t())
    C = torch.zeros(512, 512).cuda()
    hip.gemm(GEMM_NT, [(A.cuda(), B.cuda(), C)], flags=F_BF16X3)
    torch.cuda.synchronize()
    rel = ((C.cpu().double() - ref).norm() / ref.norm()).item()
    assert rel < 3e-5, rel
    C2 = torch.zeros(512, 512).cuda()
    hip.gemm(GEMM_NT, [(A.cuda(), B.cuda(), C2)])
    torch.cuda.synchronize()
    rel32 = ((C2.cpu().double() - ref).norm() / ref.norm()).item()
    assert rel32 < 1e-6, rel32
    print('relative error: bf16x3 %.3g, fp32 mfma %.3g' % (rel, rel32))


def test_gemm_asymmetric_identity(hip):
    """A = I with an asymmetric B catches a transposed C write (guide section 3)."""
    n = 96
    A = torch.eye(n)
    B = torch.arange(n * n, dtype=torch.float32).view(n, n) / 7.0
    for mode, Bm in ((GEMM_NT, B.t().contiguous()), (GEMM_NN, B), (GEMM_TN, B)):
        C = torch.zeros(n, n).cuda()
        hip.gemm(mode, [(A.cuda(), Bm.cuda(), C)])
        torch.cuda.synchronize()
        assert torch.equal(C.cpu(), B), mode


def test_gemm_strided_views_flags_groups(hip):
    def build(g):
        return dict(X=rnd(g, 70, 300), W=rnd(g, 90, 300), out=rnd(g, 4, 70, 96), big=rnd(g, 70, 200))

    def run(ops, t):
        X, W = t['X'], t['W']
        # column-sliced operands (unaligned offsets 3 and 5 -> scalar load path), accumulate + tanh into a strided C
        ops.gemm(GEMM_NT, [(X[:, 3:131], W[:, 5:133], t['big'][:, 10:100])], flags=F_ACCUM | F_TANH)
        # K-split groups into slabs
        ops.gemm(GEMM_NT, [(X[:, 0:128], W[:, 0:128], t['out'][0, :, :90]), (X[:, 128:300], W[:, 128:300], t['out'][1, :, :90])])
    both(hip, build, run, ['big', 'out'], tol=3e-5, name='gemm views')


@pytest.mark.parametrize('mode', [GEMM_NT, GEMM_NN])
@pytest.mark.parametrize('x3', [0, F_BF16X3])
@pytest.mark.parametrize('shape', [(64, 4096, 2348), (37, 100, 77), (64, 96, 1024), (5, 3072, 333), (64, 1024, 4096),
                                   (128, 4096, 2348), (100, 96, 1024), (65, 3072, 333), (128, 1024, 4096), (97, 100, 77)])
def test_gemm_skinny(hip, mode, shape, x3):
    M, N, K = shape

    def build(g):
        A = rnd(g, M, K + 4)
        B = rnd(g, N, K + 8) if mode == GEMM_NT else rnd(g, K, N + 4)
        return dict(A=A, B=B, C=rnd(g, 3, M, N + 4), bias=rnd(g, N))

    def run(ops, t):
        A = t['A'][:, 4:4 + K] if K % 4 == 0 else t['A'][:, 1:1 + K]
        if mode == GEMM_NT:
            Bv = t['B'][:, 8:8 + K] if K % 4 == 0 else t['B'][:, 3:3 + K]
        else:
            Bv = t['B'][:, 2:2 + N]
        k2 = (K // 2) // 32 * 32 or K
        Cv = [t['C'][i][:, :N] for i in range(3)]
        ops.gemm(mode, [(A, Bv, Cv[0])], alpha=0.5, bias=t['bias'], flags=F_ACCUM | F_TANH | x3)
        if k2 < K:   # K-split groups into slabs, as the engine launches the LSTM cells
            if mode == GEMM_NT:
                ops.gemm(mode, [(A[:, :k2], Bv[:, :k2], Cv[1]), (A[:, k2:], Bv[:, k2:], Cv[2])], flags=x3)
            else:
                ops.gemm(mode, [(A[:, :k2], Bv[:k2], Cv[1]), (A[:, k2:], Bv[k2:], Cv[2])], flags=x3)
    both(hip, build, run, ['C'], tol=(3e-5 if x3 else 1e-5) * max(1.0, math.sqrt(K)), name='skinny %d %s' % (mode, shape))


def test_gemm_batched(hip):
    def build(g):
        return dict(A=rnd(g, 5, 26, 64), Bm=rnd(g, 5, 40, 64), th=rnd(g, 8, 64), C1=torch.zeros(5, 26, 40),
                    C2=torch.zeros(5, 26, 8), C3=torch.zeros(5, 64, 64), P=rnd(g, 5, 26, 40))

    def run(ops, t):
        ops.gemm(GEMM_NT, [(t['A'], t['Bm'], t['C1'])], alpha=0.25)
        ops.gemm(GEMM_NT, [(t['A'], t['th'].unsqueeze(0).expand(5, 8, 64), t['C2'])])
        ops.gemm(GEMM_TN, [(t['A'], t['A'], t['C3'])])
    both(hip, build, run, ['C1', 'C2', 'C3'], name='gemm batched')


def test_gemm_large_tile_path(hip):
    def build(g):
        return dict(A=rnd(g, 1700, 520), B=rnd(g, 1030, 520), C=torch.zeros(1700, 1030), bias=rnd(g, 1030))

    def run(ops, t):
        ops.gemm(GEMM_NT, [(t['A'], t['B'], t['C'])], bias=t['bias'], flags=F_TANH)
    both(hip, build, run, ['C'], tol=2e-5, name='gemm 128 tile')


def test_slab_reduce_colsum(hip):
    def build(g):
        return dict(s=rnd(g, 5, 33, 70), out=rnd(g, 33, 80), b=rnd(g, 70), part=rnd(g, 1000, 130), cs=rnd(g, 130))

    def run(ops, t):
        ops.slab_reduce(t['s'], t['out'][:, :70], bias=t['b'], flags=F_ACCUM)
        ops.colsum(t['part'][:, 5:125], t['cs'][:120], accum=True)
    both(hip, build, run, ['out', 'cs'], tol=5e-5, name='slab/colsum')


def test_colsum_tall_chunked(hip):
    def build(g):
        return dict(part=rnd(g, 9000, 200), cs=rnd(g, 200), cs2=rnd(g, 130))

    def run(ops, t):
        ops.colsum(t['part'], t['cs'], accum=True)
        ops.colsum(t['part'][:, 7:137], t['cs2'], accum=False)
    both(hip, build, run, ['cs', 'cs2'], tol=1e-4, name='colsum tall')


@pytest.mark.parametrize('rows', [1, 63, 1664, 5000])
def test_colsum_two_destinations(hip, rows):
    def build(g):
        return dict(part=rnd(g, rows, 2048), ga=rnd(g, 1024), gb=rnd(g, 1024), b1=rnd(g, 2048), b2=rnd(g, 2048),
                    odd=rnd(g, rows, 131), o1=rnd(g, 60), o2=rnd(g, 70), o3=rnd(g, 130), o4=rnd(g, 130), w=rnd(g, 200), w2=rnd(g, 56))

    def run(ops, t):
        ops.colsum2(t['part'], t['ga'], t['gb'], split=1024, accum=True)          # LayerNorm gamma | beta partials
        ops.colsum2(t['part'], t['b1'], t['b2'], accum=False)                      # bias_ih / bias_hh
        ops.colsum2(t['odd'][:, 1:], t['o1'], t['o2'], split=60, accum=True)       # unaligned -> scalar fallback
        ops.colsum2(t['odd'][:, 1:], t['o3'], t['o4'], accum=True)
        ops.colsum2(t['part'][:, 256:512], t['w'], t['w2'], split=200, accum=False)  # aligned, uneven split
    both(hip, build, run, ['ga', 'gb', 'b1', 'b2', 'o1', 'o2', 'o3', 'o4', 'w', 'w2'], tol=1e-4, name='colsum2 %d' % rows)


def test_select_embed(hip):
    def build(g):
        lg = rnd(g, 6, 50)
        lg[2, 7] = lg[2, 30] = 40.0
        return dict(lg=lg, caps=torch.randint(4, 50, (6, 26), generator=g), coins=torch.tensor([1, 0, 1] + [0] * 23, dtype=torch.int32),
                    E=rnd(g, 50, 20), ids=torch.zeros(3, 6, dtype=torch.int64), out=torch.zeros(3, 6, 24))

    def run(ops, t):
        for k, step in enumerate((0, 1, 2)):
            ops.select_embed(t['lg'], t['caps'], step, t['coins'], t['E'], t['ids'][k], t['out'][k][:, :20], p=0.3, seed=9, site=6,
                             row0=6 * k)
    both(hip, build, run, ['ids', 'out'], tol=1e-6, name='select_embed')


@pytest.mark.parametrize('n', [48, 64, 1024, 2048, 100])
@pytest.mark.parametrize('variant', ['plain', 'tanh', 'post', 'pe_drop', 'res'])
def test_rowln_fwd_bwd(hip, n, variant):
    rows = 77
    kw = dict(plain=dict(), tanh=dict(pre_tanh=1), post=dict(post_tanh=1),
              pe_drop=dict(p1=0.3, site1=3, p2=0.2, site2=4, seed=99), res=dict(pre_tanh=1))[variant]

    def build(g):
        d = dict(x=rnd(g, rows, n + 8), gamma=1 + 0.2 * rnd(g, n), beta=0.2 * rnd(g, n), y=torch.zeros(rows, n),
                 st=torch.zeros(rows, 2), dy=rnd(g, rows, n), dx=torch.zeros(rows, n), pe=rnd(g, 26, n),
                 res=rnd(g, rows, n), part=torch.zeros(min(rows, 256), 2, n))
        return d

    def run(ops, t):
        x = t['x'][:, 4:4 + n]
        extra = dict(kw)
        if variant == 'pe_drop':
            extra['pe'] = t['pe']
        if variant == 'res':
            extra['res'] = t['res']
        ops.rowln_fwd(x, t['gamma'], t['beta'], t['y'], t['st'], **extra)
        ops.rowln_bwd(t['dy'], x, t['gamma'], t['beta'], t['dx'], stats=t['st'], dgb_part=t['part'], **extra)
        t['dgb'] = t['part'].sum(0)
    both(hip, build, run, ['y', 'st', 'dx', 'dgb'], tol=3e-5, name='rowln %d %s' % (n, variant))


def test_rowln_bwd_tall(hip):
    def build(g):
        return dict(y=torch.tanh(rnd(g, 5000, 64)), gamma=1 + 0.1 * rnd(g, 64), beta=0.1 * rnd(g, 64), dy=rnd(g, 5000, 64),
                    dx=rnd(g, 5000, 64), part=torch.zeros(1024, 2, 64), st=torch.zeros(5000, 2), o=torch.zeros(5000, 64))

    def run(ops, t):
        assert ops.rowln_bwd_nblk(5000) == 1024
        ops.rowln_fwd(t['y'], t['gamma'], t['beta'], t['o'], t['st'])
        ops.rowln_bwd(t['dy'], t['y'], t['gamma'], t['beta'], t['dx'], stats=t['st'], pre_tanh=2, dgb_part=t['part'])
        t['dgb'] = t['part'].sum(0)
    both(hip, build, run, ['dx', 'dgb'], tol=1e-4, name='rowln tall')


def test_rowln_bwd_mode2_and_accum(hip):
    def build(g):
        return dict(y=torch.tanh(rnd(g, 300, 64)), gamma=1 + 0.1 * rnd(g, 64), beta=0.1 * rnd(g, 64), dy=rnd(g, 300, 64),
                    dx=rnd(g, 300, 64), part=torch.zeros(150, 2, 64))

    def run(ops, t):
        assert ops.rowln_bwd_nblk(300) == 150 and ops.rowln_bwd_nblk(1664) == 832 and ops.rowln_bwd_nblk(4000) == 1024
        ops.rowln_bwd(t['dy'], t['y'], t['gamma'], t['beta'], t['dx'], pre_tanh=2, dgb_part=t['part'], accum_dx=True)
        t['dgb'] = t['part'].sum(0)
    both(hip, build, run, ['dx', 'dgb'], tol=3e-5, name='rowln mode2')


@pytest.mark.parametrize('shape', [(3, 26, 8), (78, 26, 1), (2, 416, 26), (4, 936, 26)])
def test_softmax_fwd_bwd(hip, shape):
    outer, n, inner = shape

    def build(g):
        return dict(x=rnd(g, outer, n, inner, scale=3.0), y=torch.zeros(outer, n, inner), dy=rnd(g, outer, n, inner),
                    dx=torch.zeros(outer, n, inner), mask=(torch.rand(outer, n, inner, generator=g) > 0.3).float())

    def run(ops, t):
        ops.softmax_fwd(t['x'], t['y'], outer, n, inner)
        ops.softmax_bwd(t['y'], t['dy'], t['dx'], outer, n, inner)
        if inner == 1:
            t['ym'] = torch.zeros_like(t['y'])
            ops.softmax_fwd(t['x'], t['ym'], outer, n, inner, mask=t['mask'])
    both(hip, build, run, ['y', 'dx'] + (['ym'] if inner == 1 else []), tol=1e-5, name='softmax %s' % (shape,))


O2V_CASES = [
    # B, T, O, H, nsplit
    (3, 26, 16, 64, 1), (3, 26, 16, 64, 4), (2, 26, 6, 64, 3), (2, 7, 5, 64, 2), (2, 32, 9, 64, 13),
    (2, 26, 16, 1024, 1), (3, 26, 16, 1024, 4), (2, 26, 36, 1024, 8), (1, 26, 16, 512, 2),
]


@pytest.mark.parametrize('case', O2V_CASES)
def test_o2v_fused(hip, case):
    B, T, O, H, ns = case
    NO = T * O

    def build(g):
        return dict(y=torch.tanh(rnd(g, B, NO, H)), v=rnd(g, B, T, H), go=1 + 0.2 * rnd(g, H), bo=0.2 * rnd(g, H),
                    z=torch.zeros(B * T, H), ml=torch.zeros(B * T, 2), os=torch.zeros(B * NO, 2), S=torch.zeros(B, NO, T))

    def run(ops, t):
        ops.o2v_fwd(t['y'], t['v'], t['go'], t['bo'], t['z'], t['ml'], t['os'], t['S'], 1.0 / math.sqrt(H / 4.0), ns)
        # the online-softmax (m, l) pair is only defined up to the split; compare the invariant log-sum-exp
        t['lse'] = t['ml'][:, 0] + torch.log(t['ml'][:, 1])
    both(hip, build, run, ['z', 'os', 'S', 'lse'], tol=3e-5, name='o2v %s' % (case,))


@pytest.mark.parametrize('case', [(3, 26, 16, 64, 2), (2, 26, 16, 1024, 1), (2, 26, 36, 1024, 3), (2, 7, 5, 512, 2)])
def test_o2v_two_graphs_in_one_launch(hip, case):
    """dlsg_o2v_fwd_multi: the object and the motion stream of CapGnnEncoder as one launch (blockIdx.z picks the argument
    block) must equal the two single launches bit for bit, and the emulation within tolerance."""
    B, T, O, H, ns = case
    NO = T * O
    g = torch.Generator().manual_seed(9)
    sc = 1.0 / math.sqrt(H / 4.0)

    def mk(dev):
        g.manual_seed(9)
        items = []
        for _ in range(2):
            items.append(dict(y=torch.tanh(rnd(g, B, NO, H)).to(dev), v=rnd(g, B, T, H).to(dev), g_obj=(1 + 0.2 * rnd(g, H)).to(dev),
                              b_obj=(0.2 * rnd(g, H)).to(dev), z=torch.zeros(B * T, H, device=dev),
                              ml=torch.zeros(B * T, 2, device=dev), ostats=torch.zeros(B * NO, 2, device=dev),
                              S=torch.zeros(B, NO, T, device=dev)))
        return items
    multi, single, emu = mk('cuda'), mk('cuda'), mk('cpu')
    hip.o2v_fwd_multi(multi, sc, ns)
    for it in single:
        hip.o2v_fwd(it['y'], it['v'], it['g_obj'], it['b_obj'], it['z'], it['ml'], it['ostats'], it['S'], sc, ns)
    EmulOps().o2v_fwd_multi(emu, sc, ns)
    for a, b_, e in zip(multi, single, emu):
        for k in ('z', 'ml', 'ostats', 'S'):
            assert torch.equal(a[k], b_[k]), k
        assert (a['z'].cpu() - e['z']).abs().max().item() <= 3e-5
        assert (a['S'].cpu() - e['S']).abs().max().item() <= 3e-5


@pytest.mark.parametrize('case', O2V_CASES)
def test_o2v_fused_backward(hip, case):
    """dlsg_o2v_bwd (scores + dv pass, apply pass, chunk combine: csrc/o2v16_bwd.hip) against the closed-form backward of the
    graph; forward state (S, ml, ostats, z) comes from each side's own forward.  dgamma | dbeta are compared after the fold
    over the (clip, chunk) partial rows."""
    B, T, O, H, ns = case
    NO = T * O

    def build(g):
        return dict(y=torch.tanh(rnd(g, B, NO, H)), v=rnd(g, B, T, H), go=1 + 0.2 * rnd(g, H), bo=0.2 * rnd(g, H),
                    z=torch.zeros(B * T, H), ml=torch.zeros(B * T, 2), os=torch.zeros(B * NO, 2), S=torch.zeros(B, NO, T),
                    dz=rnd(g, B, T, H), dy=torch.zeros(B, NO, H), dv=torch.zeros(B, T, H), part=torch.zeros(B, 2, H),
                    dysum=torch.zeros(B, H))

    def run(ops, t):
        sc = 1.0 / math.sqrt(H / 4.0)
        ops.o2v_fwd(t['y'], t['v'], t['go'], t['bo'], t['z'], t['ml'], t['os'], t['S'], sc, ns)
        dysum = torch.full((B * ns, H), float('nan'), device=t['dy'].device)
        part = ops.o2v_bwd_multi([dict(y=t['y'], ostats=t['os'], g_obj=t['go'], b_obj=t['bo'], v=t['v'], z=t['z'].view(B, T, H), dz=t['dz'],
                                       S=t['S'], ml=t['ml'], dy=t['dy'], dv=t['dv'], dysum=dysum)], sc, ns)[0]
        assert part.shape == (B * ns, 2, H)
        t['part'] = part.view(B, ns, 2, H).sum(1)
        t['dysum'] = dysum.view(B, ns, H).sum(1)      # the column sums of dy per clip (obj_embed's bias gradient before the fold)
    both(hip, build, run, ['dy', 'dv', 'part', 'dysum'], tol=5e-5, name='o2v bwd %s' % (case,))


def test_o2v_fused_backward_two_streams_one_launch(hip):
    """dlsg_o2v_bwd_multi: the object and the motion stream's graphs in one launch per pass == each stream alone."""
    B, T, O, H, ns = 5, 26, 16, 1024, 3
    NO = T * O
    g = torch.Generator().manual_seed(3)
    sc = 1.0 / math.sqrt(2048.0)
    items, single = [], []
    for i in range(2):
        t = dict(y=torch.tanh(rnd(g, B, NO, H)).cuda(), v=rnd(g, B, T, H).cuda(), g_obj=(1 + 0.2 * rnd(g, H)).cuda(),
                 b_obj=(0.2 * rnd(g, H)).cuda(), z=torch.zeros(B * T, H).cuda(), ml=torch.zeros(B * T, 2).cuda(),
                 ostats=torch.zeros(B * NO, 2).cuda(), S=torch.zeros(B, NO, T).cuda(), dz=rnd(g, B, T, H).cuda())
        hip.o2v_fwd(t['y'], t['v'], t['g_obj'], t['b_obj'], t['z'], t['ml'], t['ostats'], t['S'], sc, ns)
        t['z'] = t['z'].view(B, T, H)
        items.append(dict(t, dy=torch.zeros(B, NO, H).cuda(), dv=torch.zeros(B, T, H).cuda()))
        one = dict(t, dy=torch.zeros(B, NO, H).cuda(), dv=torch.zeros(B, T, H).cuda())
        one['part'] = hip.o2v_bwd_multi([one], sc, ns)[0]
        single.append(one)
    parts = hip.o2v_bwd_multi(items, sc, ns)
    torch.cuda.synchronize()
    for it, one, part in zip(items, single, parts):
        assert torch.equal(it['dy'], one['dy']) and torch.equal(it['dv'], one['dv']) and torch.equal(part, one['part'])


@pytest.mark.parametrize('dims', [(3, 26, 8, 64), (64, 26, 8, 1024), (2, 26, 5, 1024), (2, 32, 32, 96), (1, 7, 3, 2048),
                                  (5, 20, 17, 512)])
def test_latent_psl_fused_forward(hip, dims):
    B, T, P, H = dims

    def build(g):
        return dict(ov=rnd(g, B, T, H), th=rnd(g, P, H, scale=0.1), ga=1 + 0.2 * rnd(g, H), be=0.2 * rnd(g, H),
                    adj=torch.zeros(B, T, P), u=torch.zeros(B * P, H), out=torch.zeros(B * P, H), st=torch.zeros(B * P, 2))

    def run(ops, t):
        ops.latent_psl_fwd(t['ov'], t['th'], t['ga'], t['be'], t['adj'], t['u'], t['out'], t['st'], p=0.3, site=77, seed=9)
    both(hip, build, run, ['adj', 'u', 'out', 'st'], tol=3e-5, name='latent_psl %s' % (dims,))


@pytest.mark.parametrize('dims', [(3, 26, 128), (64, 26, 2048), (2, 32, 576), (2, 7, 64), (4, 20, 1024), (100, 26, 2048), (5, 26, 1536)])
@pytest.mark.parametrize('masked', [False, True])
def test_self_attention_core_fused_forward(hip, dims, masked):
    B, T, D = dims

    def build(g):
        return dict(K=rnd(g, B, T, D, scale=0.3), Q=rnd(g, B, T, D, scale=0.3), V=rnd(g, B, T, D),
                    mask=(torch.rand(B, T, T, generator=g) > 0.3).float(), w=torch.zeros(B, T, T), out=torch.zeros(B, T, D))

    def run(ops, t):
        ops.sa_core_fwd(t['K'], t['Q'], t['V'], t['w'], t['out'], 1.0 / math.sqrt(D / 8.0), mask=t['mask'] if masked else None)
    both(hip, build, run, ['w', 'out'], tol=3e-5, name='sa_core %s' % (dims,))


@pytest.mark.parametrize('dims', [(3, 26, 8, 64), (64, 26, 8, 1024), (2, 26, 5, 1024), (2, 32, 8, 96), (1, 7, 3, 2048), (5, 20, 1, 512)])
def test_latent_psl_fused_backward(hip, dims):
    B, T, P, H = dims

    def build(g):
        return dict(ov=rnd(g, B, T, H), th=rnd(g, P, H, scale=0.1), ga=1 + 0.2 * rnd(g, H), be=0.2 * rnd(g, H),
                    adj=torch.zeros(B, T, P), u=torch.zeros(B * P, H), out=torch.zeros(B * P, H), st=torch.zeros(B * P, 2),
                    dout=rnd(g, B * P, H), dov=torch.zeros(B * T, H), dth=torch.zeros(B, P, H), part=torch.zeros(B, 2, H))

    def run(ops, t):
        ops.latent_psl_fwd(t['ov'], t['th'], t['ga'], t['be'], t['adj'], t['u'], t['out'], t['st'], p=0.3, site=77, seed=9)
        ops.latent_psl_bwd(t['dout'], t['u'], t['st'], t['ga'], t['adj'], t['ov'], t['th'], t['dov'], t['dth'], t['part'], p=0.3,
                           site=77, seed=9)
    both(hip, build, run, ['dov', 'dth', 'part'], tol=3e-5, name='latent_psl bwd %s' % (dims,))


@pytest.mark.parametrize('dims', [(3, 26, 128), (64, 26, 2048), (2, 32, 576), (2, 7, 64), (4, 20, 1024), (100, 26, 2048), (5, 26, 1536)])
def test_self_attention_core_fused_backward(hip, dims):
    B, T, D = dims

    def build(g):
        return dict(K=rnd(g, B, T, D, scale=0.3), Q=rnd(g, B, T, D, scale=0.3), V=rnd(g, B, T, D), dout=rnd(g, B, T, D),
                    w=torch.zeros(B, T, T), out=torch.zeros(B, T, D), dK=torch.zeros(B, T, D), dQ=torch.zeros(B, T, D),
                    dV=torch.zeros(B, T, D))

    def run(ops, t):
        sc = 1.0 / math.sqrt(D / 8.0)
        ops.sa_core_fwd(t['K'], t['Q'], t['V'], t['w'], t['out'], sc)
        ops.sa_core_bwd(t['w'], t['K'], t['Q'], t['V'], t['dout'], t['dK'], t['dQ'], t['dV'], sc)
    both(hip, build, run, ['dK', 'dQ', 'dV'], tol=3e-5, name='sa_core bwd %s' % (dims,))


def test_o2v_online_softmax_rescale_branch(hip):
    """Force the running max to jump at a late tile (guide 5.4 rule 26): one object aligned with one frame."""
    B, T, O, H = 2, 26, 16, 64
    NO = T * O
    g = torch.Generator().manual_seed(5)
    y = torch.tanh(rnd(g, B, NO, H)) * 0.1
    v = rnd(g, B, T, H)
    y[:, NO - 3] = torch.tanh(v[:, 11] * 3.0)          # huge score for frame 11 in the last tile
    y[:, 40] = torch.tanh(v[:, 5] * 3.0)
    go, bo = torch.ones(H), torch.zeros(H)
    outs = {}
    for nm, ops, dev in (('c', EmulOps(), 'cpu'), ('g', hip, 'cuda')):
        z = torch.zeros(B * T, H, device=dev); ml = torch.zeros(B * T, 2, device=dev)
        os_ = torch.zeros(B * NO, 2, device=dev); S = torch.zeros(B, NO, T, device=dev)
        ops.o2v_fwd(y.to(dev), v.to(dev), go.to(dev), bo.to(dev), z, ml, os_, S, 1.0, 1)
        outs[nm] = z.cpu()
    assert torch.isfinite(outs['g']).all()
    assert (outs['c'] - outs['g']).abs().max().item() <= 5e-5


@pytest.mark.parametrize('dims', [(5, 8, 48, 64, 2), (64, 8, 1024, 1024, 2), (3, 26, 48, 64, 1), (7, 5, 1024, 1024, 2)])
def test_decatt_fwd_bwd(hip, dims):
    B, P, Q, H, ns = dims

    def build(g):
        d = dict(q=rnd(g, B, Q + 4), alpha=torch.zeros(B, ns * P), dalpha=rnd(g, B, ns * P), dq=rnd(g, B, Q + 8))
        for s in range(ns):
            d['K%d' % s] = rnd(g, B, P, Q, scale=0.2); d['V%d' % s] = rnd(g, B, P, H)
            d['c%d' % s] = torch.zeros(B, H); d['dc%d' % s] = rnd(g, B, H)
            d['dK%d' % s] = rnd(g, B, P, Q); d['dV%d' % s] = rnd(g, B, P, H)
        return d

    def run(ops, t):
        Kp = [t['K%d' % s] for s in range(ns)]; Vp = [t['V%d' % s] for s in range(ns)]
        q = t['q'][:, 2:2 + Q]
        ops.decatt_fwd(Kp, Vp, q, [t['c%d' % s] for s in range(ns)], t['alpha'], 0.3)
        ops.decatt_bwd(Kp, Vp, q, t['alpha'], [t['dc%d' % s] for s in range(ns)], [t['dK%d' % s] for s in range(ns)],
                       [t['dV%d' % s] for s in range(ns)], t['dq'][:, 4:4 + Q], 0.3, accum_dq=True, dalpha=t['dalpha'])
    outs = ['alpha', 'dq'] + [k % s for s in range(ns) for k in ('c%d', 'dK%d', 'dV%d')]
    both(hip, build, run, outs, tol=3e-5, name='decatt %s' % (dims,))


@pytest.mark.parametrize('dims', [(3, 48), (64, 1024), (5, 96)])
def test_lstm_pointwise(hip, dims):
    B, H = dims

    def build(g):
        return dict(slabs=rnd(g, 3, B, 4 * H), add=rnd(g, B, 2, 4 * H), bi=rnd(g, 4 * H), bh=rnd(g, 4 * H), cp=rnd(g, B, H),
                    c=torch.zeros(B, H), h=torch.zeros(B, 2 * H), h2=torch.zeros(B, H), gates=torch.zeros(B, 3, 4 * H),
                    dh=rnd(g, B, H), dh2=rnd(g, B, H), dcn=rnd(g, B, H), dg=torch.zeros(B, 4 * H), dcp=torch.zeros(B, H))

    def run(ops, t):
        gates = t['gates'][:, 1]
        ops.lstm_pw_fwd(t['slabs'], t['c'], B, H, addend=t['add'][:, 1], b_ih=t['bi'], b_hh=t['bh'], c_prev=t['cp'],
                        h=t['h'][:, H:], h2=t['h2'], gates=gates, p=0.3, site=7, seed=5)
        ops.lstm_pw_bwd(gates, t['c'], t['dg'], B, H, c_prev=t['cp'], dh=t['dh'], dh2=t['dh2'], dc_next=t['dcn'],
                        dc_prev=t['dcp'], p=0.3, site=7, seed=5)
        t['c0'] = torch.zeros_like(t['c']); t['g0'] = torch.zeros(B, 4 * H, device=t['c'].device)
        ops.lstm_pw_fwd(None, t['c0'], B, H, addend=t['add'][:, 0], gates=t['g0'])
    both(hip, build, run, ['c', 'h', 'h2', 'gates', 'dg', 'dcp', 'c0', 'g0'], tol=1e-5, name='lstm_pw %s' % (dims,))


@pytest.mark.parametrize('dims', [(3, 48), (64, 1024), (7, 100)])
def test_lstm_pointwise_two_directions_one_launch(hip, dims):
    B, H = dims

    def build(g):
        d = {}
        for k in range(2):
            d.update({'slabs%d' % k: rnd(g, 3 + k, B, 4 * H), 'add%d' % k: rnd(g, B, 5, 4 * H), 'bi%d' % k: rnd(g, 4 * H),
                      'bh%d' % k: rnd(g, 4 * H), 'cp%d' % k: rnd(g, B, H), 'c%d' % k: torch.zeros(B, H),
                      'h%d' % k: torch.zeros(B, 2 * H), 'h2%d' % k: torch.zeros(B, H), 'gates%d' % k: torch.zeros(B, 4 * H),
                      'dh%d' % k: rnd(g, B, H), 'rec%d' % k: rnd(g, 4, B, 2 * H), 'dcn%d' % k: rnd(g, B, H),
                      'dg%d' % k: torch.zeros(B, 4 * H), 'dcp%d' % k: torch.zeros(B, H)})
        return d

    def run(ops, t):
        ops.lstm_pw_fwd_multi([dict(slabs=t['slabs%d' % k], c=t['c%d' % k], B=B, H=H, addend=t['add%d' % k][:, 2 + k],
                                    b_ih=t['bi%d' % k], b_hh=t['bh%d' % k], c_prev=t['cp%d' % k] if k else None,
                                    h=t['h%d' % k][:, k * H:(k + 1) * H], h2=t['h2%d' % k], gates=t['gates%d' % k])
                               for k in range(2)])
        ops.lstm_pw_bwd_multi([dict(gates=t['gates%d' % k], c=t['c%d' % k], dgates=t['dg%d' % k], B=B, H=H,
                                    c_prev=t['cp%d' % k] if k else None, dh=t['dh%d' % k],
                                    dh4=t['rec%d' % k][:, :, k * H:(k + 1) * H], dc_next=t['dcn%d' % k], dc_prev=t['dcp%d' % k])
                               for k in range(2)])
    outs = [n + str(k) for k in range(2) for n in ('c', 'h', 'h2', 'gates', 'dg', 'dcp')]
    both(hip, build, run, outs, tol=1e-5, name='lstm_pw pair %s' % (dims,))


@pytest.mark.parametrize('dims', [(3, 48, 32, 40, 5, 2), (64, 1024, 1024, 1024, 8, 2), (5, 96, 64, 80, 3, 1), (2, 600, 520, 300, 32, 2)])
def test_dec_step_fused_fwd(hip, dims):
    """dec_mid_fwd / dec_tail_fwd against the unfused chain they replace (the emulator composes the unfused ops)."""
    B, Q, H, D, P, ns = dims

    def build(g):
        d = dict(slabs=rnd(g, 3, B, 4 * Q), add=rnd(g, B, 2, 4 * Q), bi=rnd(g, 4 * Q), bh=rnd(g, 4 * Q), cp=rnd(g, B, Q),
                 c=torch.zeros(B, Q), h=torch.zeros(B, Q), gates=torch.zeros(B, 4 * Q), gq=rnd(g, Q), bq=rnd(g, Q),
                 qcur=torch.zeros(B, Q), stq=torch.zeros(B, 2), alpha=torch.zeros(B, ns * P),
                 slabs2=rnd(g, 2, B, 4 * D), bi2=rnd(g, 4 * D), bh2=rnd(g, 4 * D), cp2=rnd(g, B, D), c2=torch.zeros(B, D),
                 hd=torch.zeros(B, D), gates2=torch.zeros(B, 4 * D), gl=rnd(g, D), bl=rnd(g, D), dout=torch.zeros(B, D),
                 stl=torch.zeros(B, 2))
        for s in range(ns):
            d['K%d' % s] = rnd(g, B, P, Q, scale=0.2); d['V%d' % s] = rnd(g, B, P, H)
            d['g%d' % s] = rnd(g, H); d['b%d' % s] = rnd(g, H)
            d['cpre%d' % s] = torch.zeros(B, H); d['ctx%d' % s] = torch.zeros(B, H); d['stc%d' % s] = torch.zeros(B, 2)
        return d

    def run(ops, t):
        R = range(ns)
        ops.dec_mid_fwd(t['slabs'], t['add'][:, 1], t['bi'], t['bh'], t['cp'], t['c'], t['h'], t['gates'], (t['gq'], t['bq']),
                        t['qcur'], t['stq'], 0.3, 11, [t['K%d' % s] for s in R], [t['V%d' % s] for s in R],
                        [(t['g%d' % s], t['b%d' % s]) for s in R], [t['cpre%d' % s] for s in R], [t['ctx%d' % s] for s in R],
                        [t['stc%d' % s] for s in R], t['alpha'], [0.2, 0.4][:ns], [21, 22][:ns], 0.3, seed=5)
        ops.dec_tail_fwd(t['slabs2'], t['bi2'], t['bh2'], t['cp2'], t['c2'], t['hd'], t['gates2'], (t['gl'], t['bl']),
                         t['dout'], t['stl'], 0.3, 31, seed=5)
    outs = ['c', 'h', 'gates', 'qcur', 'stq', 'alpha', 'c2', 'hd', 'gates2', 'dout', 'stl']
    outs += [k % s for s in range(ns) for k in ('cpre%d', 'ctx%d', 'stc%d')]
    both(hip, build, run, outs, tol=3e-5, name='dec_step %s' % (dims,))


@pytest.mark.parametrize('dims', [(3, 40, 57, 12), (64, 1024, 1000, 300), (5, 80, 130, 33), (7, 1024, 2048, 300)])
@pytest.mark.parametrize('coin', [0, 1])
def test_dec_tail_samples_the_next_word(hip, dims, coin):
    """dec_tail_fwd(sample=...): on a step whose coin is 0 the launch projects every row onto the vocabulary, takes the first
    maximum and writes the id and its (word-dropped) embedding for the next step -- what dlsg_gemm + dlsg_select_embed did in two
    more launches; with the coin set (teacher-forced) it leaves the prefilled id and embedding alone.  ids bit-exact (the
    vocabulary rows are well separated here), embedding rows equal."""
    B, D, V, W = dims

    def build(g):
        Wv = rnd(g, V, D)
        d = dict(slabs2=rnd(g, 2, B, 4 * D), bi2=rnd(g, 4 * D), bh2=rnd(g, 4 * D), cp2=rnd(g, B, D), c2=torch.zeros(B, D),
                 hd=torch.zeros(B, D), gates2=torch.zeros(B, 4 * D), gl=rnd(g, D), bl=rnd(g, D), dout=torch.zeros(B, D),
                 stl=torch.zeros(B, 2), Wv=Wv, bv=rnd(g, V), E=rnd(g, V, W), ids=torch.full((B,), 3, dtype=torch.int64),
                 we=torch.full((B, W + 4), 7.0), coins=torch.tensor([1, coin, 1], dtype=torch.int32))
        return d

    def run(ops, t):
        ops.dec_tail_fwd(t['slabs2'], t['bi2'], t['bh2'], t['cp2'], t['c2'], t['hd'], t['gates2'], (t['gl'], t['bl']),
                         t['dout'], t['stl'], 0.3, 31, seed=5,
                         sample=dict(coins=t['coins'], t=1, W=t['Wv'], b=t['bv'], E=t['E'], ids_out=t['ids'], we_out=t['we'][:, :W],
                                     p=0.25, site=9, row0=2 * B))
    both(hip, build, run, ['c2', 'hd', 'gates2', 'dout', 'stl', 'ids', 'we'], tol=3e-5, name='dec_tail sample %s coin %d' % (dims, coin))


@pytest.mark.parametrize('dims', [(3, 48, 32, 40, 5, 2, 3), (64, 1024, 1024, 1024, 8, 2, 5), (5, 96, 64, 80, 3, 1, 1),
                                  (2, 600, 520, 300, 32, 2, 6)])
@pytest.mark.parametrize('last', [False, True])
def test_dec_step_fused_bwd(hip, dims, last):
    """dec_mid_bwd (+ slab-stack dh4 of lstm_pw_bwd, + decatt_cache_grads) against the unfused backward chain.
    last=True: the final word step (no recurrent inputs); write_rec is exercised by last=False."""
    B, Q, H, D, P, ns, S = dims
    L = 4

    def build(g):
        d = dict(slabs=rnd(g, S, B, ns * H + Q + D), rec=rnd(g, 3, B, Q + D), dlh=torch.zeros(B, D), alpha=torch.softmax(rnd(g, B, ns, P), 2).reshape(B, ns * P),
                 dalpha=rnd(g, B, ns * P), ds=torch.zeros(B, ns * P), qh=rnd(g, B, Q), gq=rnd(g, Q), partq=torch.zeros(B, 2, Q),
                 gates=torch.sigmoid(rnd(g, B, 4 * Q)), c=rnd(g, B, Q), cp=rnd(g, B, Q), dc=rnd(g, B, Q), dg=torch.zeros(B, 4 * Q),
                 gl=torch.sigmoid(rnd(g, B, 4 * D)), lc=rnd(g, B, D), lcp=rnd(g, B, D), dlho=rnd(g, B, D), dlc=rnd(g, B, D),
                 dgl=torch.zeros(B, 4 * D), dlh3=rnd(g, B, D),
                 A=torch.softmax(rnd(g, L, B, ns, P), 3).reshape(L, B, ns * P), DS=rnd(g, L, B, ns * P), QC=rnd(g, L, B, Q))
        qh = d['qh']
        d['stq'] = torch.cat([qh.mean(1, keepdim=True), 1 / torch.sqrt(qh.var(1, unbiased=False, keepdim=True) + 1e-5)], 1)
        for s in range(ns):
            d['K%d' % s] = rnd(g, B, P, Q, scale=0.2); d['V%d' % s] = rnd(g, B, P, H)
            d['g%d' % s] = rnd(g, H); d['cpre%d' % s] = rnd(g, B, H)
            y = torch.tanh(d['cpre%d' % s])
            d['stc%d' % s] = torch.cat([y.mean(1, keepdim=True), 1 / torch.sqrt(y.var(1, unbiased=False, keepdim=True) + 1e-5)], 1)
            d['partc%d' % s] = torch.zeros(B, 2, H); d['dcpre%d' % s] = torch.zeros(B, H)
            d['DC%d' % s] = rnd(g, L, B, H); d['dK%d' % s] = torch.zeros(B, P, Q); d['dV%d' % s] = torch.zeros(B, P, H)
        return d

    def run(ops, t):
        R = range(ns)
        rec = None if last else t['rec']
        ops.lstm_pw_bwd(t['gl'], t['lc'], t['dgl'], B, D, c_prev=t['lcp'], dh2=t['dlho'], dh3=None if last else t['dlh3'],
                        dh4=None if last else rec[:, :, Q:], dc_next=t['dlc'], dc_prev=t['dlc'], p=0.3, site=9, seed=5)
        ops.dec_mid_bwd(t['slabs'], None if last else t['dlh'], [t['cpre%d' % s] for s in R], [t['stc%d' % s] for s in R],
                        [t['g%d' % s] for s in R], [t['partc%d' % s] for s in R], [t['dcpre%d' % s] for s in R], [0.2, 0.4][:ns],
                        [21, 22][:ns], [t['K%d' % s] for s in R], [t['V%d' % s] for s in R], t['alpha'],
                        t['dalpha'] if last else None, t['ds'], t['qh'], t['stq'], t['gq'], t['partq'], 0.3, 11,
                        None if last else rec[:, :, :Q], t['gates'], t['c'], t['cp'], t['dc'], t['dg'], 0.3, seed=5)
        ops.decatt_cache_grads(t['A'], t['DS'], t['QC'], [t['DC%d' % s] for s in R], [t['dK%d' % s] for s in R],
                               [t['dV%d' % s] for s in R])
    outs = ['dgl', 'dlc', 'ds', 'partq', 'dc', 'dg'] + ([] if last else ['dlh'])
    outs += [k % s for s in range(ns) for k in ('partc%d', 'dcpre%d', 'dK%d', 'dV%d')]
    both(hip, build, run, outs, tol=3e-5, name='dec_step_bwd %s' % (dims,))


@pytest.mark.parametrize('dims', [(7, 5, 1000), (128, 5, 1000), (3, 3, 50), (4, 8, 10007), (2, 1, 40)])
@pytest.mark.parametrize('first', [False, True])
def test_beam_select_and_state_gather(hip, dims, first):
    B, k, V = dims
    R = B * k
    end = 2

    def build(g):
        last = torch.randint(3, V, (R,), generator=g)
        last[torch.rand(R, generator=g) < 0.4] = end               # finished beams, including whole groups
        last[:k] = end
        return dict(lg=rnd(g, R, V + 3, scale=3.0), last=last, lp=-torch.rand(R, generator=g) * 5,
                    pred=torch.zeros(R, dtype=torch.int64), nlp=torch.zeros(R), back=torch.zeros(R, dtype=torch.int64),
                    rows=torch.zeros(R, dtype=torch.int64), cnt=torch.zeros(1, dtype=torch.int32),
                    s0=rnd(g, R, 64), s1=rnd(g, R, 100), s2=rnd(g, R, 1024), s3=rnd(g, R, 7),
                    d0=torch.zeros(R, 64), d1=torch.zeros(R, 100), d2=torch.zeros(R, 1024), d3=torch.zeros(R, 7))

    def run(ops, t):
        ops.beam_select(t['lg'][:, 1:V + 1], t['last'], t['lp'], t['pred'], t['nlp'], t['back'], t['rows'], k, end, first=first,
                        ended_count=t['cnt'])
        ops.gather_rows_multi([t['s%d' % i] for i in range(4)], t['rows'], [t['d%d' % i] for i in range(4)])
    both(hip, build, run, ['pred', 'nlp', 'back', 'rows', 'cnt', 'd0', 'd1', 'd2', 'd3'], tol=1e-5, name='beam_select %s' % (dims,))


@pytest.mark.parametrize('rows,n,pad', [(5, 4096 * 3, 0), (7, 4100, 4), (3, 1001, 0), (64, 26 * 6144, 0), (9, 20, 3), (1, 8191, 1)])
def test_gather_rows_bit_exact(hip, rows, n, pad):
    """row pieces of 4096 floats per workgroup: whole pieces, a ragged last piece, unaligned rows (scalar path), strided views"""
    g = torch.Generator().manual_seed(rows * 1000 + n)
    src = torch.randn(23, n + pad, generator=g).cuda()
    idx = torch.randint(0, 23, (rows,), generator=g).cuda()
    dst = torch.full((rows, n + pad), -7.0).cuda()
    hip.gather_rows(src[:, :n], idx, dst[:, :n])
    torch.cuda.synchronize()
    assert torch.equal(dst[:, :n], src[idx][:, :n])
    assert (dst[:, n:] == -7.0).all()


def test_movers_embed_argmax(hip):
    def build(g):
        lg = rnd(g, 9, 1000)
        lg[3, 17] = lg[3, 900] = 50.0           # exact tie -> lowest index
        lg[5, 999] = 60.0
        return dict(x=rnd(g, 4, 8, 64), out=torch.zeros(4, 200), dout=rnd(g, 4, 200), dx=rnd(g, 4, 8, 64),
                    E=rnd(g, 50, 20), ids=torch.randint(0, 50, (30,), generator=g), we=torch.zeros(30, 24),
                    dwe=rnd(g, 30, 24), dE=torch.zeros(50, 20), lg=lg, am=torch.zeros(9, dtype=torch.int64),
                    src=rnd(g, 26, 5, 33), dst=torch.zeros(5, 26, 33), dr=rnd(g, 40, 64), dro=torch.zeros(40, 64),
                    f=rnd(g, 1000))

    def run(ops, t):
        ops.mean_rows_fwd(t['x'], t['out'][:, 100:164])
        ops.mean_rows_bwd(t['dout'][:, 3:67], t['dx'], accum=True)
        ops.embed_fwd(t['E'], t['ids'], t['we'][:, :20], p=0.3, seed=11, site=6, row0=60)
        ops.embed_bwd(t['dwe'][:, 2:22], t['ids'], t['dE'], p=0.3, seed=11, site=6, row0=60)
        ops.argmax(t['lg'], t['am'])
        ops.permute_tb(t['src'], t['dst'])
        ops.dropout(t['dr'], t['dro'], 0.3, 77, 5)
        ops.copy2d(t['dr'][:, :32], t['dro'][:, 32:], accum=True)
        ops.fill(t['f'], 2.5)
    both(hip, build, run, ['out', 'dx', 'we', 'dE', 'am', 'dst', 'dro', 'f'], tol=1e-5, name='movers')


def test_argmax_of_rows_without_a_maximum_is_word_zero(hip):
    """a logit row of all NaN / all -inf has no maximum: the kernels answer 0 (torch.argmax's answer for -inf rows) and gather
    row 0 of the embedding -- never an id outside the vocabulary (the id indexes E in the same launch)"""
    lg = torch.randn(4, 300, device='cuda')
    lg[1] = float('nan')
    lg[2] = float('-inf')
    am = torch.full((4,), -1, dtype=torch.int64, device='cuda')
    hip.argmax(lg, am)
    want = lg.argmax(1)
    assert am[0] == want[0] and am[3] == want[3] and am[1] == 0 and am[2] == 0
    E = torch.randn(300, 20, device='cuda')
    ids = torch.full((4,), -1, dtype=torch.int64, device='cuda')
    out = torch.zeros(4, 20, device='cuda')
    hip.select_embed(lg, torch.zeros(4, 26, dtype=torch.int64, device='cuda'), 0, torch.zeros(26, dtype=torch.int32, device='cuda'), E, ids, out)
    torch.cuda.synchronize()
    assert torch.equal(ids, am) and torch.equal(out, E[am])


@pytest.mark.parametrize('tm', [True, False])
def test_ce_ragged_and_log_softmax(hip, tm):
    B, L, V = 5, 26, 61

    def build(g):
        lg = rnd(g, L, B, V, scale=2.0) if tm else rnd(g, B, L, V, scale=2.0)
        return dict(lg=lg, tgt=torch.randint(0, V, (B, L), generator=g), lens=torch.tensor([26, 19, 30, 5, 1]),
                    dl=torch.zeros_like(lg), rl=torch.zeros(B * L), loss=torch.zeros(1), ls=torch.zeros(B * L, V))

    def run(ops, t):
        ops.ce_ragged(t['lg'], t['tgt'], t['lens'], t['dl'], t['rl'], t['loss'], tm)
        ops.log_softmax(t['lg'].view(B * L, V), t['ls'])
    both(hip, build, run, ['dl', 'loss', 'ls'], tol=1e-5, name='ce')


def test_adam_matches_torch_optim(hip):
    g = torch.Generator().manual_seed(3)
    p0, grads = rnd(g, 5000), [rnd(g, 5000) for _ in range(3)]
    ref = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([ref], lr=1.6e-4, betas=(0.5, 0.9))
    p = p0.clone().cuda(); m = torch.zeros_like(p); v = torch.zeros_like(p)
    for i, gr in enumerate(grads):
        ref.grad = gr.clone() / 4
        opt.step()
        hip.adam(p, gr.cuda(), m, v, 1.6e-4, 0.5, 0.9, 1e-8, i + 1, 0.25)
    torch.cuda.synchronize()
    assert (p.cpu() - ref.detach()).abs().max().item() <= 1e-6


@pytest.mark.parametrize('off,n', [(0, 5000), (1, 4999), (3, 1026), (2, 3), (5, 1), (0, 4), (7, 100003)])
def test_adam_ranges_at_any_offset(hip, off, n):
    """trainable ranges of the parameter arena start at arbitrary parameter boundaries: scalar head / 16-byte body / scalar tail"""
    g = torch.Generator().manual_seed(off * 100 + n)
    tot = off + n + 9
    p0, gr = rnd(g, tot), rnd(g, tot)
    m0, v0 = rnd(g, tot).abs() * 0.1, rnd(g, tot).abs() * 0.1
    p, gg, m, v = p0.clone().cuda(), gr.cuda(), m0.clone().cuda(), v0.clone().cuda()
    hip.adam(p[off:off + n], gg[off:off + n], m[off:off + n], v[off:off + n], 1.6e-4, 0.5, 0.9, 1e-8, 3, 0.5)
    torch.cuda.synchronize()
    from emul_ops import EmulOps
    pe, me, ve = p0.clone(), m0.clone(), v0.clone()
    EmulOps().adam(pe[off:off + n], gr[off:off + n], me[off:off + n], ve[off:off + n], 1.6e-4, 0.5, 0.9, 1e-8, 3, 0.5)
    for a, b in ((p, pe), (m, me), (v, ve)):
        assert (a.cpu() - b).abs().max().item() <= 1e-6
        assert torch.equal(a.cpu()[:off], b[:off]) and torch.equal(a.cpu()[off + n:], b[off + n:])        # nothing outside the range


def test_masked_self_attention_core_against_reference_fixture(hip):
    """tests/golden/sa_mask.npz (the reference's SelfAttention with an attention mask, sublayer.py:70-72, as DiscV2 uses it):
    PE add and the bias-free projections in torch, the masked 26 x 26 core on the HIP kernel."""
    import os
    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'sa_mask.npz')))
    x = torch.from_numpy(g['x']).cuda()
    B, T, D = x.shape
    x = x + torch.from_numpy(g['w.pe.pe'])[:, :T].cuda()              # the fixture's module was built with get_pe=True
    Dp = 64                                                            # the kernel takes widths that are multiples of 64:
    K, Q, V = (torch.nn.functional.pad(x @ torch.from_numpy(g['w.%s.weight' % n]).cuda().t(), (0, Dp - D)).contiguous()
               for n in 'KQV')                                         # zero columns change neither the logits nor out[:, :, :D]
    assert hip.sa_core_supported(T, Dp)
    for mask, key in ((torch.from_numpy(g['mask']).cuda().contiguous(), 'y_masked'), (None, 'y')):
        w = torch.empty(B, T, T, device='cuda'); out = torch.empty(B, T, Dp, device='cuda')
        hip.sa_core_fwd(K, Q, V, w, out, 1.0 / math.sqrt(D), mask=mask)
        y = out[:, :, :D] @ torch.from_numpy(g['w.output_layer.0.weight']).cuda().t()
        assert np.abs(y.cpu().numpy() - g[key]).max() <= 2e-5, key


@pytest.mark.parametrize('mode', ['nt', 'nn', 'tn'])
def test_large_gemm_launches_of_the_step(hip, mode):
    """Launches of several thousand 128 x 128 tiles in the step's own grouping (two groups sharing A with bias + tanh; 16
    row-chunk groups writing slabs, i.e. the group-per-XCD tile map; accumulate epilogue) against torch on the GPU."""
    from dlsg_amd.hip import GEMM_NT, GEMM_NN, GEMM_TN, F_TANH, F_ACCUM
    g = torch.Generator(device='cuda').manual_seed(1)

    def r(*s):
        return torch.randn(*s, device='cuda', generator=g)
    if mode == 'nt':        # the step's region projection in miniature: 2 groups share A; 3328 tiles of 128 x 128
        M, N, K = 26624, 1024, 256
        A = r(M, K); Ws = [r(N, K) * 0.1, r(N, K) * 0.1]; bs = [r(N), r(N)]
        Cs = [torch.empty(M, N, device='cuda') for _ in range(2)]
        hip.gemm(GEMM_NT, [(A, W, C, b) for W, C, b in zip(Ws, Cs, bs)], flags=F_TANH)
        for W, C, b in zip(Ws, Cs, bs):
            ref = torch.tanh(A @ W.t() + b)
            assert (C - ref).abs().max().item() <= 2e-4
    elif mode == 'nn':
        M, N, K = 13312, 2048, 192          # 104 x 16 = 1664 tiles = 2.17 rounds
        A = r(M, K); W = r(K, N) * 0.1
        C = r(M, N)
        ref = C + A @ W
        hip.gemm(GEMM_NN, [(A, W, C)], flags=F_ACCUM)
        assert (C - ref).abs().max().item() <= 5e-4
    else:                   # deep weight gradient in miniature: 16 row-chunk groups -> slabs, 8 x 16 tiles each = 2048 tiles
        Mo, No, rows = 1024, 2048, 16 * 160
        dY = r(rows, Mo) * 0.1; X = r(rows, No)
        slabs = torch.empty(16, Mo, No, device='cuda')
        hip.gemm(GEMM_TN, [(dY[i * 160:(i + 1) * 160], X[i * 160:(i + 1) * 160], slabs[i]) for i in range(16)])
        ref = dY.t() @ X
        assert (slabs.sum(0) - ref).abs().max().item() <= 1e-3
        for i in (0, 7, 15):
            assert (slabs[i] - dY[i * 160:(i + 1) * 160].t() @ X[i * 160:(i + 1) * 160]).abs().max().item() <= 5e-4


def test_colsum_tall_is_bit_reproducible(hip):
    """26 624-row column sums (the region projection's bias gradient): chunk partials are combined in a fixed order"""
    g = torch.Generator().manual_seed(3)
    part = (torch.randn(26624, 1024, generator=g) * 3).cuda()
    outs = []
    for _ in range(12):
        a, b = torch.zeros(1024, device='cuda'), torch.full((1024,), 0.5, device='cuda')
        hip.colsum(part, a)
        hip.colsum2(part, b, a.clone(), accum=True)
        outs.append((a.clone(), b.clone()))
    torch.cuda.synchronize()
    assert all(torch.equal(outs[0][0], o[0]) and torch.equal(outs[0][1], o[1]) for o in outs)
    ref = part.double().sum(0)
    assert (outs[0][0].double() - ref).abs().max().item() <= 2e-3


@pytest.mark.parametrize('shape,dim', [((192, 26, 3), 1), ((192, 26, 26), 2), ((192, 26, 1), 1), ((192, 2), 1), ((5, 130, 7), 1)])
def test_softmax_forward_and_backward(hip, shape, dim):
    """softmax forward / backward over an inner or the last axis"""
    outer = int(np.prod(shape[:dim])); n = shape[dim]; inner = int(np.prod(shape[dim + 1:]))

    def build(g):
        return dict(x=rnd(g, *shape, scale=2.0), dy=rnd(g, *shape), y=torch.zeros(*shape), dx=torch.zeros(*shape))

    def run(ops, t):
        ops.softmax_fwd(t['x'], t['y'], outer, n, inner)
        ops.softmax_bwd(t['y'], t['dy'], t['dx'], outer, n, inner)
    both(hip, build, run, ['y', 'dx'], tol=2e-5, name='softmax %s' % (shape,))


def test_embed_bwd_with_a_dominant_id(hip):
    """the word-embedding gradient when one id (the <pad> word of the caption tails) owns 40 % of 1664 rows: fixed row order,
    bit-reproducible, equal to the emulation"""
    def build(g):
        ids = torch.randint(1, 1000, (1664,), generator=g)
        ids[torch.rand(1664, generator=g) < 0.4] = 0
        return dict(dwe=rnd(g, 1664, 300), ids=ids, dE=torch.zeros(1000, 300), dE2=torch.zeros(1000, 300))

    def run(ops, t):
        ops.embed_bwd(t['dwe'], t['ids'], t['dE'], p=0.3, seed=5, site=2, row0=0)
        ops.embed_bwd(t['dwe'], t['ids'], t['dE2'], p=0.0, seed=0, site=0, row0=0)
    both(hip, build, run, ['dE', 'dE2'], tol=2e-5, name='embed_bwd dominant id')
    g = torch.Generator().manual_seed(9)
    ids = torch.randint(0, 5, (1664,), generator=g).cuda()
    dwe = torch.randn(1664, 300, generator=g).cuda()
    outs = []
    for _ in range(5):
        dE = torch.zeros(1000, 300, device='cuda')
        hip.embed_bwd(dwe, ids, dE, p=0.0, seed=0, site=0, row0=0)
        outs.append(dE)
    assert all(torch.equal(outs[0], o) for o in outs)


# ------------------------------------------------------------------------------------------------ persistent BiLSTM recurrence
def _bilstm_reference(xg, Whh, bih, bhh, B, T, H):
    """plain torch fp32 restatement of nn.LSTM's recurrence (gate order i, f, g, o; models/layer.py:26,52) given the x-gates"""
    out = torch.zeros(B, T, 2 * H)
    hprev = [torch.zeros(B, T, H), torch.zeros(B, T, H)]
    cst = [torch.zeros(B, T, H), torch.zeros(B, T, H)]
    gates = [torch.zeros(B, T, 4 * H), torch.zeros(B, T, 4 * H)]
    for d in range(2):
        h = torch.zeros(B, H); c = torch.zeros(B, H)
        order = range(T) if d == 0 else range(T - 1, -1, -1)
        for t in order:
            hprev[d][:, t] = h
            pre = xg[d].view(B, T, 4 * H)[:, t] + bih[d] + bhh[d] + h @ Whh[d].t()
            i, f, g, o = pre.split(H, dim=1)
            i, f, g, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(g), torch.sigmoid(o)
            c = f * c + i * g
            h = o * torch.tanh(c)
            out[:, t, d * H:(d + 1) * H] = h
            cst[d][:, t] = c
            gates[d][:, t] = torch.cat([i, f, g, o], 1)
    return out, hprev, cst, gates


@pytest.mark.parametrize('B,T,H', [(3, 26, 64), (64, 26, 64), (1, 1, 64), (2, 7, 512), (64, 26, 1024), (17, 26, 1024), (64, 40, 512)])
def test_persistent_bilstm_forward(hip, B, T, H):
    """csrc/bilstm.hip: all T steps of both directions in ONE launch (W_hh slices resident in LDS, h_t exchanged between the
    workgroups through L2 with write-through stores and flags) against the plain recurrence; twice, so that stale exchange
    data or flags of the first launch would show; the time-out word must stay 0."""
    assert hip.lib.dlsg_bilstm_supported(B, T, H) == 1
    assert hip.lib.dlsg_bilstm_supported(65, T, H) == 0 and hip.lib.dlsg_bilstm_supported(B, T, 96) == 0
    g = torch.Generator().manual_seed(7)
    sc = 1.0 / math.sqrt(H)
    xg = [rnd(g, B * T, 4 * H), rnd(g, B * T, 4 * H)]
    Whh = [rnd(g, 4 * H, H, scale=2 * sc), rnd(g, 4 * H, H, scale=2 * sc)]
    bih = [rnd(g, 4 * H, scale=0.3), rnd(g, 4 * H, scale=0.3)]
    bhh = [rnd(g, 4 * H, scale=0.3), rnd(g, 4 * H, scale=0.3)]
    want = _bilstm_reference(xg, Whh, bih, bhh, B, T, H)
    cu = lambda ts: [t.cuda() for t in ts]
    xg_, W_, bi_, bh_ = cu(xg), cu(Whh), cu(bih), cu(bhh)
    for rep in range(2):
        out = torch.full((B, T, 2 * H), float('nan'), device='cuda')
        hprev = [torch.zeros(B, T, H, device='cuda') for _ in range(2)]
        cst = [torch.full((B, T, H), float('nan'), device='cuda') for _ in range(2)]
        gates = [torch.full((B, T, 4 * H), float('nan'), device='cuda') for _ in range(2)]
        err = hip.bilstm_fwd(xg_, W_, bi_, bh_, out, hprev, cst, gates)
        torch.cuda.synchronize()
        assert int(err.item()) == 0, 'a workgroup timed out waiting for its producers'
        got = (out, hprev, cst, gates)
        for name, a, b in (('out', want[0], got[0]), ('hprev0', want[1][0], got[1][0]), ('hprev1', want[1][1], got[1][1]),
                           ('c0', want[2][0], got[2][0]), ('c1', want[2][1], got[2][1]), ('gates0', want[3][0], got[3][0]),
                           ('gates1', want[3][1], got[3][1])):
            e = (a - b.cpu()).abs().max().item()
            assert e <= 3e-5, (rep, name, e)


@pytest.mark.parametrize('B,T,H', [(3, 26, 64), (64, 26, 64), (1, 1, 64), (2, 7, 512), (64, 26, 1024), (17, 26, 1024)])
def test_persistent_bilstm_backward(hip, B, T, H):
    """backward through time in one launch against torch autograd of the plain recurrence: d loss / d (gate pre-activations)
    of every step and direction for a random d loss / d out; run twice (reused exchange buffers / flags)."""
    g = torch.Generator().manual_seed(11)
    sc = 1.0 / math.sqrt(H)
    xg = [rnd(g, B * T, 4 * H).requires_grad_(True) for _ in range(2)]
    Whh = [rnd(g, 4 * H, H, scale=2 * sc), rnd(g, 4 * H, H, scale=2 * sc)]
    bih = [rnd(g, 4 * H, scale=0.3), rnd(g, 4 * H, scale=0.3)]
    bhh = [rnd(g, 4 * H, scale=0.3), rnd(g, 4 * H, scale=0.3)]
    dout = rnd(g, B, T, 2 * H)
    # autograd through the recurrence: the gradient w.r.t. xg IS the gradient w.r.t. the gate pre-activations
    outs = []
    for d in range(2):
        h = torch.zeros(B, H); c = torch.zeros(B, H)
        hs = [None] * T
        for t in (range(T) if d == 0 else range(T - 1, -1, -1)):
            pre = xg[d].view(B, T, 4 * H)[:, t] + bih[d] + bhh[d] + h @ Whh[d].t()
            i, f, gg, o = pre.split(H, dim=1)
            c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
            h = torch.sigmoid(o) * torch.tanh(c)
            hs[t] = h
        outs.append(torch.stack(hs, 1))
    out = torch.cat(outs, 2)
    (out * dout).sum().backward()
    want = [xg[d].grad.view(B, T, 4 * H) for d in range(2)]
    with torch.no_grad():
        _, _, cst, gates = _bilstm_reference([x.detach() for x in xg], Whh, bih, bhh, B, T, H)
    cu = lambda ts: [t.contiguous().cuda() for t in ts]
    gates_, c_, W_ = cu(gates), cu(cst), cu(Whh)
    dout_ = dout.cuda()
    for rep in range(2):
        dG = [torch.full((B, T, 4 * H), float('nan'), device='cuda') for _ in range(2)]
        err = hip.bilstm_bwd(gates_, c_, dout_, W_, dG)
        torch.cuda.synchronize()
        assert int(err.item()) == 0, 'a workgroup timed out waiting for its producers (code %d)' % int(err.item())
        for d in range(2):
            ref = want[d]
            e = (ref - dG[d].cpu()).abs().max().item()
            assert e <= 2e-5 * max(1.0, ref.abs().max().item()), (rep, d, e, ref.abs().max().item())


def test_grouped_column_sums_equal_the_single_launches(hip):
    """dlsg_colsum_multi: the short column sums of a backward collected into grouped launches -- bit-identical to launching each
    on its own (same additions, same order), including two sums that accumulate into the same destination (kept in order, in
    different launches), the two-destination forms, strided inputs and an input too tall for the grouped kernel."""
    g = torch.Generator().manual_seed(5)
    cases = []
    for rows, n, kind in ((37, 128, 'one'), (1664, 4096, 'dup'), (200, 2048, 'split'), (64, 64, 'one'), (3000, 1024, 'one'),
                          (5000, 256, 'one'), (333, 512, 'strided'), (90, 128, 'same_dest'), (91, 128, 'same_dest')):
        part = rnd(g, rows, n + (64 if kind == 'strided' else 0)).cuda()
        cases.append((part[:, :n] if kind == 'strided' else part, n, kind))

    def run(defer):
        outs = []
        shared = torch.zeros(128, device='cuda')
        if defer:
            hip.colsum_defer = []
        for part, n, kind in cases:
            if kind == 'dup':
                a, b = torch.ones(n, device='cuda'), torch.ones(n, device='cuda')
                hip.colsum2(part, a, b, accum=True); outs += [a, b]
            elif kind == 'split':
                a, b = torch.zeros(n // 2, device='cuda'), torch.zeros(n // 2, device='cuda')
                hip.colsum2(part, a, b, split=n // 2); outs += [a, b]
            elif kind == 'same_dest':
                hip.colsum(part, shared, accum=True)
            else:
                a = torch.full((n,), 0.5, device='cuda')
                hip.colsum(part, a, accum=(kind == 'strided')); outs.append(a)
        if defer:
            assert len(hip.colsum_defer) == len(cases) - 1          # the 5000-row input went out on its own
            hip.colsum_flush()
            assert hip.colsum_defer is None
        torch.cuda.synchronize()
        return outs + [shared]
    single, grouped = run(False), run(True)
    for a, b in zip(single, grouped):
        assert torch.equal(a, b)
    ref = cases[1][0].double().sum(0).float().cpu() + 1.0
    assert (single[1].cpu() - ref).abs().max().item() <= 1e-3


def _plain_lstm(xin, W):
    """nn.LSTM's recurrence with the input products already in xin (L, n, 4H): plain torch, differentiable"""
    L, n, G = xin.shape
    H = G // 4
    h = xin.new_zeros(n, H); c = xin.new_zeros(n, H)
    hs = []
    for t in range(L):
        i, f, g, o = (xin[t] + h @ W.t()).split(H, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        hs.append(h)
    return torch.stack(hs)


def _gp_like(hs_of, xin, W, r1, r2):
    """a scalar that differentiates the LSTM twice, as the critic's gradient penalty does (run_gun.py:362-371)"""
    hs = hs_of(xin, W)
    g, = torch.autograd.grad((hs * r1).sum(), xin, create_graph=True)
    pen = ((g.reshape(g.shape[0], g.shape[1], -1).norm(dim=-1) - 1.0) ** 2).mean()
    return pen * 10.0 + (hs * r2).sum() * 0.01



def test_adam_is_guarded_by_the_persistent_time_out_word(hip):
    """a persistent recurrent launch whose hand-off timed out leaves invalid gradients: dlsg_adam reads the same device word and
    updates nothing; check_persistent() raises, clears the word and switches the BiLSTM to its step-by-step schedule"""
    p = torch.ones(1000, device='cuda')
    g = torch.full((1000,), 0.5, device='cuda')
    m, v = torch.zeros(1000, device='cuda'), torch.zeros(1000, device='cuda')
    word = hip._persist_word(p.device)
    try:
        word.fill_(7)
        hip.adam(p, g, m, v, 1e-2, 0.5, 0.9, 1e-8, 1)
        torch.cuda.synchronize()
        assert torch.equal(p, torch.ones_like(p)) and float(m.abs().max()) == 0.0
        with pytest.raises(RuntimeError):
            hip.check_persistent()
        assert int(word.item()) == 0 and hip.persistent_bilstm is False
        hip.check_persistent()
        hip.adam(p, g, m, v, 1e-2, 0.5, 0.9, 1e-8, 1)
        torch.cuda.synchronize()
        assert float((p - 1).abs().max()) > 5e-3
    finally:
        word.zero_()
        hip.persistent_bilstm = True
