"""CPU emulation of the attention fragment pipeline (DESIGN.md "fragment layout").

The kernels never shuffle data between lanes: the weight-row permutations chosen at pack time make the
MFMA accumulators of the QKV GEMM *be* the operand fragments of the attention MFMAs.  This test replays
that index algebra in numpy with the documented gfx950 32x32x16 lane maps
(A: row = lane&31, k = 8*(lane>>5)+j;  B: col = lane&31, same k;  C/D: col = lane&31,
row = (r&3) + 8*(r>>2) + 4*(lane>>5)) using the REAL permutation tables exported by the library, and
checks the result against a plain attention computed from the un-permuted weights."""
import ctypes

import numpy as np


def mfma(a, b, c):
    """a, b: [64 lanes][8]; c: [64][16] -> d [64][16] (v_mfma_f32_32x32x16 semantics)."""
    A = np.zeros((32, 16)); B = np.zeros((16, 32))
    for l in range(64):
        A[l & 31, 8 * (l >> 5):8 * (l >> 5) + 8] = a[l]
        B[8 * (l >> 5):8 * (l >> 5) + 8, l & 31] = b[l]
    D = A @ B
    d = c.copy()
    for l in range(64):
        for r in range(16):
            d[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return d


def panel_frag(X, tile, ks):
    return np.stack([X[tile * 32 + (l & 31), ks * 16 + (l >> 5) * 8: ks * 16 + (l >> 5) * 8 + 8] for l in range(64)])


def wfrag(W, rowmap, ft, ks):
    return np.stack([W[rowmap[ft * 32 + (l & 31)], ks * 16 + (l >> 5) * 8: ks * 16 + (l >> 5) * 8 + 8] for l in range(64)])


def maps():
    import mdgen_amd._lib as L
    arrs = [(ctypes.c_int32 * 384)() for _ in range(5)]
    assert L.lib.mdgen_debug_layout_maps(*arrs) == 0
    return [np.array(a) for a in arrs]


def test_maps_are_permutations_with_rotary_pairs():
    qk, vf, vs, pqk, pvs = maps()
    for m in (qk, vf, vs, pqk, pvs):
        assert sorted(m.tolist()) == list(range(384))
    # lane-order slots (2p, 2p+1) of half h are the rotary pair (6h+p, 6h+p+12) of the same head
    for i in range(0, 384, 2):
        e = i % 12
        h = (i // 48) % 2
        assert pqk[i + 1] == pqk[i] + 12 and pqk[i] % 24 == 6 * h + e // 2


def test_fragment_pipeline_reproduces_attention():
    qk, vf, vs, pqk, pvs = maps()
    rng = np.random.default_rng(0)
    C, KS = 384, 24
    X = rng.standard_normal((64, C))
    Wq, Wk, Wv = (rng.standard_normal((C, C)) / 20 for _ in range(3))
    w = 2                                   # wave 2: heads 8..11
    def gemm_T(W, rowmap):                  # acc[ft][tt] = D[feature][token]
        acc = [[np.zeros((64, 16)) for _ in range(2)] for _ in range(3)]
        for ks in range(KS):
            for ft in range(3):
                for tt in range(2):
                    acc[ft][tt] = mfma(wfrag(W, rowmap, 3 * w + ft, ks), panel_frag(X, tt, ks), acc[ft][tt])
        return acc
    def heads_T(acc):                       # epilogue_heads_T without bias/rope -> frag[hd][tt] = (ks0 [64][8], ks1 [64][8])
        out = {}
        for tt in range(2):
            for hd in range(4):
                e = np.zeros((64, 12))
                for c in range(3):
                    ap = 3 * hd + c
                    ft, a = ap >> 2, ap & 3
                    for b in range(4):
                        e[:, 4 * c + b] = acc[ft][tt][:, 4 * a + b]
                k1 = np.zeros((64, 8)); k1[:, :4] = e[:, 8:12]
                out[hd, tt] = (e[:, :8].copy(), k1)
        return out
    qf, kf = heads_T(gemm_T(Wq, qk)), heads_T(gemm_T(Wk, qk))
    accv = [[np.zeros((64, 16)) for _ in range(3)] for _ in range(2)]   # V non-transposed: acc[tt][j]
    for ks in range(KS):
        for tt in range(2):
            for j in range(3):
                accv[tt][j] = mfma(panel_frag(X, tt, ks), wfrag(Wv, vf, 3 * w + j, ks), accv[tt][j])
    vfrag = {}                              # epilogue_v_flash: [hd][tt][ks][half][d] = 8 values
    for j in range(3):
        for l in range(64):
            col = 32 * j + (l & 31)
            hd, d, hh = col // 24, col % 24, l >> 5
            for tt in range(2):
                vfrag[hd, tt, 0, hh, d] = accv[tt][j][l, :8].copy()
                vfrag[hd, tt, 1, hh, d] = accv[tt][j][l, 8:].copy()
    q, k, v = X @ Wq.T, X @ Wk.T, X @ Wv.T
    for hd in range(4):
        head = 4 * w + hd
        sl = slice(head * 24, head * 24 + 24)
        for qt in range(2):
            o_ref = np.zeros((32, 24))
            O = np.zeros((64, 16))
            for kt in range(2):
                S = mfma(kf[hd, kt][0], qf[hd, qt][0], np.zeros((64, 16)))
                S = mfma(kf[hd, kt][1], qf[hd, qt][1], S)
                v0 = np.zeros((64, 8)); v1 = np.zeros((64, 8))
                for l in range(64):
                    if (l & 31) < 24:
                        v0[l] = vfrag[hd, kt, 0, l >> 5, l & 31]
                        v1[l] = vfrag[hd, kt, 1, l >> 5, l & 31]
                O = mfma(v0, S[:, :8], O)
                O = mfma(v1, S[:, 8:], O)
                s_ref = q[qt * 32:(qt + 1) * 32, sl] @ k[kt * 32:(kt + 1) * 32, sl].T
                o_ref += s_ref @ v[kt * 32:(kt + 1) * 32, sl]
            got = np.zeros((32, 24))
            for l in range(64):
                got[l & 31, (l >> 5) * 12:(l >> 5) * 12 + 12] = O[l, :12]       # flash epilogue store
                assert np.allclose(O[l, 12:], 0)
            assert np.allclose(got, o_ref, rtol=1e-9, atol=1e-9), (hd, qt)
    # SMALL layout: per-token slots [head][half][12]; q.k over the 24 slots == true dot product,
    # V slots are the natural feature order
    accs = gemm_T(Wv, vs)
    vsm = heads_T(accs)
    for hd in range(4):
        head = 4 * w + hd
        for tt in range(2):
            for l in range(64):
                tok, hh = tt * 32 + (l & 31), l >> 5
                val = np.concatenate([vsm[hd, tt][0][l], vsm[hd, tt][1][l, :4]])
                assert np.allclose(val, v[tok, head * 24 + 12 * hh: head * 24 + 12 * hh + 12])
                qq = np.concatenate([qf[hd, tt][0][l], qf[hd, tt][1][l, :4]])
                feats = pqk[((w * 2 + hh) * 4 + hd) * 12: ((w * 2 + hh) * 4 + hd) * 12 + 12]
                assert np.allclose(qq, q[tok, feats])


# ---- row-owner MLP kernel (csrc/k_rows.hip, rows.h) ---------------------------------------------------------------
def kappa(ks, hh, j):
    """K order of every row-owner weight fragment (rows.h): element j of lane half hh in k-step ks."""
    return 16 * ks + 8 * (j >> 2) + 4 * hh + (j & 3)


def stream_frag(W, tile, ks):
    """A-operand fragment [64 lanes][8] of weight rows 32 tile .. + 31, K slice of k-step ks in kappa order
    (what k_pack_stream writes)."""
    return np.stack([[W[tile * 32 + (l & 31), kappa(ks, l >> 5, j)] for j in range(8)] for l in range(64)])


def mlp_stream_table():
    import mdgen_amd._lib as L
    buf = (ctypes.c_int32 * 2304)()
    assert L.lib.mdgen_debug_mlp_stream_table(buf, 2304) == 2304
    return np.array(buf)


def test_mlp_stream_table_covers_every_fragment_once():
    t = mlp_stream_table()
    mat, tile, ks = t >> 16, (t >> 8) & 255, t & 255
    fc1 = set(zip(tile[mat == 0].tolist(), ks[mat == 0].tolist()))
    fc2 = set(zip(tile[mat == 1].tolist(), ks[mat == 1].tolist()))
    assert len(fc1) == 48 * 24 == (mat == 0).sum() and fc1 == {(a, b) for a in range(48) for b in range(24)}
    assert len(fc2) == 12 * 96 == (mat == 1).sum() and fc2 == {(a, b) for a in range(12) for b in range(96)}


def test_row_owner_mlp_pipeline_reproduces_the_mlp():
    """Replays k_mlp_rows for one wave (32 tokens) with the library's stream table: fragments are consumed strictly in
    stream order by the kernel's block schedule (P0, P1, 22 pipelined iterations, E0, E1); the fc1 accumulators start
    from the bias (re-armed per GELU group), GELU output registers 8s .. 8s + 7 are the B operand of k-step s of fc2, and
    the residual epilogue reads feature 32 ft + 8 a + 4 hh + i from accumulator register 4 a + i."""
    tab = mlp_stream_table()
    rng = np.random.default_rng(1)
    C, F = 384, 1536
    X = rng.standard_normal((32, C))                 # LN'd + modulated rows of the wave
    W1 = rng.standard_normal((F, C)) / 20
    W2 = rng.standard_normal((C, F)) / 40
    b1 = rng.standard_normal(F) / 4
    act = np.tanh                                   # any elementwise function (the kernel's is the erf GELU)
    mats = {0: W1, 1: W2}
    frags = iter(range(2304))

    def next_frag(expect_mat):
        f = next(frags)
        e = int(tab[f])
        assert e >> 16 == expect_mat
        return stream_frag(mats[e >> 16], (e >> 8) & 255, e & 255), (e >> 8) & 255, e & 255

    # B-operand fragments of the rows: lane (n, hh), element j = feature kappa(ks, hh, j)  (rows.h rows_ln)
    xf = [np.stack([[X[l & 31, kappa(ks, l >> 5, j)] for j in range(8)] for l in range(64)]) for ks in range(24)]
    hh = np.arange(64) >> 5

    def arm(c):   # accumulators of chunk c start from its bias: register r of tile t <- b1[64 c + 32 t + row(r, hh)]
        a = np.zeros((2, 64, 16))
        for t in range(2):
            for r in range(16):
                a[t, :, r] = b1[64 * c + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hh] if c < 24 else 0.0
        return a

    a1 = [arm(0), arm(1)]
    hf = [np.zeros((4, 64, 8)), np.zeros((4, 64, 8))]
    y = np.zeros((12, 64, 16))

    def xblock(c, kx, buf):
        for q in range(12):
            w, tile, ks = next_frag(0)
            assert (tile, ks) == (2 * c + (q & 1), 6 * kx + (q >> 1))
            a1[buf][q & 1] = mfma(w, xf[ks], a1[buf][q & 1])

    def yblock(c, kk, buf):
        for q in range(12):
            w, tile, ks = next_frag(1)
            assert (tile, ks) == (q, 4 * c + kk)
            y[q] = mfma(w, hf[buf][kk], y[q])

    def gelu_group(c, g, abuf, hbuf, rearm_chunk):
        tile, a = g >> 2, g & 3
        kk, e0 = 2 * tile + (a >> 1), 4 * (a & 1)
        for j in range(4):
            hf[hbuf][kk][:, e0 + j] = act(a1[abuf][tile][:, 4 * a + j])
            if rearm_chunk is not None and rearm_chunk < 24:
                a1[abuf][tile][:, 4 * a + j] = b1[64 * rearm_chunk + 32 * tile + 8 * a + 4 * hh + j]

    for kx in range(4):
        xblock(0, kx, 0)                                      # P0
    for kx in range(4):                                       # P1: X(1) with GELU(0), a1[0] re-armed for chunk 2
        xblock(1, kx, 1)
        gelu_group(0, 2 * kx, 0, 0, 2)
        gelu_group(0, 2 * kx + 1, 0, 0, 2)
    for c in range(1, 23):                                    # X(c+1) -> a1[(c+1)&1], GELU(c): a1[c&1] -> hf[c&1], Y(c-1) <- hf[(c-1)&1]
        for b in range(8):
            if b % 2 == 0:
                xblock(c + 1, b // 2, (c + 1) & 1)
            else:
                yblock(c - 1, b // 2, (c - 1) & 1)
            gelu_group(c, b, c & 1, c & 1, c + 2)
    for kk in range(4):                                       # E0: Y(22) with GELU(23)
        yblock(22, kk, 0)
        gelu_group(23, 2 * kk, 1, 1, None)
        gelu_group(23, 2 * kk + 1, 1, 1, None)
    for kk in range(4):
        yblock(23, kk, 1)                                     # E1
    assert next(frags, None) is None
    # epilogue addressing: token n = lane & 31, feature 32 ft + 8 a + 4 hh + i <- y[ft][lane][4 a + i]
    out = np.zeros((32, C))
    for l in range(64):
        for ft in range(12):
            for a in range(4):
                for i in range(4):
                    out[l & 31, 32 * ft + 8 * a + 4 * (l >> 5) + i] = y[ft][l, 4 * a + i]
    ref = act(X @ W1.T + b1) @ W2.T
    np.testing.assert_allclose(out, ref, rtol=1e-9, atol=1e-9)


def test_training_attention_workgroup_order_keeps_a_sequence_on_one_xcd():
    """csrc/k_attn16.hip `wg_of` / `wg_grid`: workgroup b runs on XCD b % 8 (hardware round robin, one L2 per XCD).  The mapping
    must (a) cover every (sequence, head, row block) exactly once, (b) send all 16 x nblk workgroups of a sequence to ONE XCD
    (each of them streams the whole sequence's rows: with consecutive workgroups per sequence the PMC counters showed 1.3 GiB read
    per launch for 0.5 GiB of operands), (c) pad the grid only with workgroups whose sequence index is past the end (they exit),
    and (d) keep the eight XCDs equally loaded up to that padding.  Replayed here in Python for the shapes of the trunk's two
    axes at ATLAS size, a tetrapeptide axis and sequence counts that are not multiples of eight."""
    kH = 16

    def wg_of(b, nblk):
        xcd, rest, per_seq = b & 7, b >> 3, nblk * kH
        within = rest % per_seq
        return (rest // per_seq) * 8 + xcd, within // nblk, within % nblk

    def wg_grid(nseq, nblk):
        return ((nseq + 7) // 8) * 8 * kH * nblk

    for nseq, nblk in ((250, 2), (256, 2), (256, 3), (64, 8), (5, 1), (9, 3), (1, 1)):
        grid = wg_grid(nseq, nblk)
        seen, xcd_of_seq, load = set(), {}, [0] * 8
        for b in range(grid):
            seq, hd, blk = wg_of(b, nblk)
            assert 0 <= hd < kH and 0 <= blk < nblk
            if seq >= nseq:
                continue                                   # padding workgroup: exits before any barrier
            assert (seq, hd, blk) not in seen
            seen.add((seq, hd, blk))
            assert xcd_of_seq.setdefault(seq, b & 7) == (b & 7)
            load[b & 7] += 1
        assert len(seen) == nseq * kH * nblk
        assert max(load) - min(load) <= kH * nblk          # at most one sequence of difference between XCDs


def test_small_split_block_mapping_covers_every_panel_and_chunk_once():
    """csrc/k_gemm.hip `k_mlp8<PRE, S>` (option `small_split`) and `k_ln_qkv8<true>`, index algebra replayed in Python.
    k_mlp8: block b -> (xcd = b & 7, s = (b >> 3) % S, panel = ((b >> 3) / S) * 8 + xcd); grid = ceil(panels / 8) * 8 * S.  The S
    workgroups of a panel must (a) share `b % 8` (= one XCD, one L2: their fp32 partials meet there), (b) cover the twelve hidden
    chunks exactly once between their 2 S wave groups (group g of workgroup s: chunks NC (2 s + g) .. + NC - 1, NC = 12 / (2 S)),
    (c) own disjoint slots of the partial / private-row scratch (slot = panel * S + s, < panels * S), and padding blocks (panel >=
    panels) must be whole workgroups that exit.  k_ln_qkv8<true>: block b -> (panel = b >> 1, half = b & 1): q and k on half 0
    (wave groups 0 / 1), v on half 1 (group 0), each product exactly once per panel."""
    kNChunk, S = 12, 3
    NC = kNChunk // (2 * S)
    for panels in (1, 2, 7, 8, 9, 63, 85, 96):
        grid = (panels + 7) // 8 * 8 * S
        seen, xcd_of_panel, slots = {}, {}, set()
        for b in range(grid):
            xcd, rest = b & 7, b >> 3
            s, pn = rest % S, (rest // S) * 8 + xcd
            if pn >= panels:
                continue                                   # padding: the whole workgroup returns before its first barrier
            assert xcd_of_panel.setdefault(pn, xcd) == xcd
            slot = pn * S + s
            assert slot not in slots and slot < panels * S
            slots.add(slot)
            for g in range(2):
                c0 = NC * (2 * s + g)
                for c in range(c0, c0 + NC):
                    assert (pn, c) not in seen
                    seen[(pn, c)] = (s, g)
        assert len(seen) == panels * kNChunk and len(slots) == panels * S
    for panels in (1, 5, 63, 128):
        done = {}
        for b in range(2 * panels):
            pn, half = b >> 1, b & 1
            for g in range(2):
                do_q, do_k, do_v = half == 0 and g == 0, half == 0 and g == 1, half == 1 and g == 0
                for name, on in (("q", do_q), ("k", do_k), ("v", do_v)):
                    if on:
                        assert (pn, name) not in done
                        done[(pn, name)] = (half, g)
        assert len(done) == 3 * panels
