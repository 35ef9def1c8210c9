"""CPU check of the packed-weight layout (csrc/field_layout.h + field_pack.cpp).

The blob built from the library's gather table is consumed here by a lane-exact model of
v_mfma_f32_32x32x2_f32 (operand and accumulator lane maps as documented for gfx950) that walks
the weight stream exactly the way the fused kernel does: activations as the B operand straight
from the accumulator registers, weights as the A operand from 1 KiB chunks.  The result must equal
the oracle's field network.  This pins the layout logic on CPU; the kernel itself is checked on
the GPU (tests/test_gpu_parity.py)."""
import numpy as np
import pytest
import torch

import cases
from conftest import rel_err
from oracle import dmnerf_cpu as oracle

import dm_nerf_amd

CHUNK, STAGE, CONST = 256, 8192, 3840
STAGE_L0, STAGE_L1, STAGE_L5, STAGE_L6 = 0, 2, 34, 44
STAGE_RGBF, STAGE_RGBH, STAGE_INSF, STAGE_INSH, STAGE_INSO = 60, 68, 73, 81, 85
C_BIAS8, C_BIAS4, C_BIASO, C_DENSW, C_DENSB, C_RGBW, C_RGBB = 0, 2560, 2816, 2944, 3200, 3204, 3588

LANE = np.arange(64)
P, HALF = LANE & 31, LANE >> 5


def feat_of(i, h):
    return 32 * (i >> 4) + (i & 3) + 8 * ((i & 15) >> 2) + 4 * h


def pos_slot(j, h, n_freq_slots=30):
    if j == 0:
        return 0 if h == 0 else 2
    if j == 1:
        return 1 if h == 0 else -1
    if j - 2 >= n_freq_slots:
        return -1
    return 3 + 6 * ((j - 2) // 3) + 3 * h + (j - 2) % 3


class Wave:
    """32 points on one wavefront: accumulators as dense [tile][32 features][32 points]."""

    def __init__(self, blob):
        self.cst = blob[:CONST]
        self.stream = blob[CONST:]

    def chunk(self, stage, c):
        base = (stage * 32 + c) * CHUNK
        return self.stream[base:base + CHUNK].reshape(64, 4)          # [lane][e]

    @staticmethod
    def mfma(acc_tile, a_lane, b_lane):
        # A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]  ->  D[i][j] += sum_k A[i][k] B[k][j]
        A = np.zeros((32, 2)); B = np.zeros((2, 32))
        A[P, HALF] = a_lane
        B[HALF, P] = b_lane
        acc_tile += A @ B

    @staticmethod
    def regs(acc):
        """accumulator tiles -> register file view [i][lane] (C/D map of the 32x32 MFMA)."""
        nt = acc.shape[0]
        out = np.zeros((16 * nt, 64))
        for i in range(16 * nt):
            r = i & 15
            out[i] = acc[i >> 4, (r & 3) + 8 * (r >> 2) + 4 * HALF, P]
        return out

    def bias(self, off, nt):
        acc = np.zeros((nt, 32, 32))
        for i in range(16 * nt):
            r = i & 15
            for h in range(2):
                acc[i >> 4, (r & 3) + 8 * (r >> 2) + 4 * h, :] = self.cst[off + h * 16 * nt + i]
        return acc

    def layer8(self, acc, stage0, kstep0, src_regs):
        for j, b in enumerate(src_regs):
            k = kstep0 + j
            for q in range(2):
                ch = self.chunk(stage0 + (2 * k + q) // 32, (2 * k + q) % 32)
                for e in range(4):
                    self.mfma(acc[4 * q + e], ch[:, e], b)

    def layer4(self, acc, stage0, kstep0, src_regs):
        for j, b in enumerate(src_regs):
            k = kstep0 + j
            ch = self.chunk(stage0 + k // 32, k % 32)
            for e in range(4):
                self.mfma(acc[e], ch[:, e], b)


def simulate(blob, emb, ins_num):
    """emb: [32, 90] encodings of the wave's 32 points -> raw [32, 5 + ins_num]."""
    T = (ins_num + 1 + 31) // 32
    w = Wave(blob.astype(np.float64))
    pe = np.zeros((32, 64)); qe = np.zeros((16, 64))
    for j in range(32):
        for h in range(2):
            s = pos_slot(j, h)
            if s >= 0:
                pe[j, HALF == h] = emb[:, s]
    for j in range(16):
        for h in range(2):
            s = pos_slot(j, h, n_freq_slots=12)
            if s >= 0:
                qe[j, HALF == h] = emb[:, 63 + s]
    acc = w.bias(C_BIAS8 + 0 * 256, 8)
    w.layer8(acc, STAGE_L0, 0, pe)
    act = np.maximum(w.regs(acc), 0)
    for l in range(1, 8):
        stage0 = STAGE_L1 + 8 * (l - 1) if l <= 4 else (STAGE_L5 if l == 5 else STAGE_L6 + 8 * (l - 6))
        acc = w.bias(C_BIAS8 + l * 256, 8)
        w.layer8(acc, stage0, 0, act)
        if l == 5:
            w.layer8(acc, stage0, 128, pe)
        act = np.maximum(w.regs(acc), 0)
    # density head on the VALU
    sig = np.zeros(64)
    for i in range(128):
        sig += act[i] * w.cst[C_DENSW + HALF * 128 + i]
    sigma = sig[:32] + sig[32:] + w.cst[C_DENSB]
    # colour branch
    acc = w.bias(C_BIAS8 + 8 * 256, 8)
    w.layer8(acc, STAGE_RGBF, 0, act)
    g = w.regs(acc)
    acc2 = w.bias(C_BIAS4, 4)
    w.layer4(acc2, STAGE_RGBH, 0, g)
    w.layer4(acc2, STAGE_RGBH, 128, qe)
    r = np.maximum(w.regs(acc2), 0)
    rgb = np.zeros((3, 32))
    for c in range(3):
        v = np.zeros(64)
        for i in range(64):
            v += r[i] * w.cst[C_RGBW + c * 128 + HALF * 64 + i]
        rgb[c] = v[:32] + v[32:] + w.cst[C_RGBB + c]
    # instance branch
    acc = w.bias(C_BIAS8 + 9 * 256, 8)
    w.layer8(acc, STAGE_INSF, 0, act)
    e = w.regs(acc)
    acc2 = w.bias(C_BIAS4 + 128, 4)
    w.layer4(acc2, STAGE_INSH, 0, e)
    s = np.maximum(w.regs(acc2), 0)
    acc3 = np.zeros((T, 32, 32))
    for i in range(16 * T):
        rr = i & 15
        for h in range(2):
            acc3[i >> 4, (rr & 3) + 8 * (rr >> 2) + 4 * h, :] = w.cst[C_BIASO + h * 64 + i]
    JS = 16 if T <= 2 else 8
    for jg in range(16):
        for t in range(T):
            ch = w.chunk(STAGE_INSO + jg // JS, (jg % JS) * T + t)
            for ee in range(4):
                w.mfma(acc3[t], ch[:, ee], s[4 * jg + ee])
    ins = acc3.reshape(T * 32, 32)[: ins_num + 1]
    return np.concatenate([rgb, sigma[None], ins], 0).T


def build_blob(ins_num, params):
    lib = dm_nerf_amd.load_library()
    n = lib.dmn_field_blob_floats(ins_num)
    idx = np.empty(n, dtype=np.int32)
    assert lib.dmn_field_pack_index(ins_num, idx.ctypes.data) == 0
    flat = np.concatenate([params[k].numpy().reshape(-1) for k, _ in oracle.param_specs(ins_num)])
    assert flat.size == lib.dmn_field_param_floats(ins_num)
    blob = np.where(idx >= 0, flat[np.clip(idx, 0, None)], 0.0).astype(np.float32)
    return blob, idx, flat


@pytest.mark.parametrize("ins_num,seed", [(13, 0), (93, 3), (40, 5)])
def test_blob_consumed_in_mfma_order_equals_oracle_field(ins_num, seed):
    params = oracle.make_params(ins_num, seed)
    blob, idx, flat = build_blob(ins_num, params)
    x, v = cases.field_rows()
    emb = torch.cat([oracle.encode(torch.from_numpy(x[:32]), 10), oracle.encode(torch.from_numpy(v[:32]), 4)], -1)
    with torch.no_grad():
        want = oracle.field(params, emb).numpy()
    got = simulate(blob, emb.numpy().astype(np.float64), ins_num)
    assert got.shape == want.shape
    assert rel_err(got, want) < 5e-6


def test_every_parameter_is_packed_exactly_once_per_use():
    for ins_num in (13, 93):
        lib = dm_nerf_amd.load_library()
        n = lib.dmn_field_blob_floats(ins_num)
        idx = np.empty(n, dtype=np.int32)
        assert lib.dmn_field_pack_index(ins_num, idx.ctypes.data) == 0
        total = lib.dmn_field_param_floats(ins_num)
        used = idx[idx >= 0]
        assert used.max() == total - 1
        counts = np.bincount(used, minlength=total)
        assert counts.min() == 1 and counts.max() == 1      # each weight / bias lands in exactly one slot
        assert n == CONST + (85 + (1 if ins_num + 1 <= 64 else 2)) * STAGE


def test_layout_rejects_unsupported_instance_counts():
    lib = dm_nerf_amd.load_library()
    assert lib.dmn_field_blob_floats(128) < 0          # ins_num + 1 = 129 > 128
    assert lib.dmn_field_blob_floats(127) > 0
    assert lib.dmn_field_param_floats(-1) < 0


# ---------------------------------------------------------------------------------------------
# split-precision ("x3") blob: lane-exact model of v_mfma_f32_32x32x16_{f16,bf16} operand maps
# (A: lane l holds A[l&31][8*(l>>5) + j], B: B[8*(l>>5) + j][l&31], j = 0..7), tiles in the outer loop
# ---------------------------------------------------------------------------------------------
X3_STAGE = {"l0": 0, "l1": 1, "l5": 17, "l6": 22, "rgbf": 30, "rgbh": 34, "insf": 37, "insh": 41, "inso": 43}
X3_CHUNK = 512


class WaveX3:
    def __init__(self, cst, stream):
        self.cst, self.stream = cst, stream

    def layer(self, stage0, n_tiles, frags, bias_off):
        """frags: list of k-step operand fragments [8][64] (element j, lane).  Returns tiles [n_tiles][32][32]."""
        nk = len(frags)
        acc = np.zeros((n_tiles, 32, 32))
        for t in range(n_tiles):
            for i in range(16):
                for h in range(2):
                    acc[t, (i & 3) + 8 * (i >> 2) + 4 * h, :] = self.cst[bias_off + h * 16 * n_tiles + 16 * t + i]
            for ks in range(nk):
                chunk = stage0 * 64 + (t * nk + ks) * 2          # hi chunk (the lo chunk is zero in this model)
                a = self.stream[chunk * X3_CHUNK:(chunk + 1) * X3_CHUNK].reshape(64, 8)
                A = np.zeros((32, 16)); B = np.zeros((16, 32))
                for j in range(8):
                    A[P, 8 * HALF + j] = a[:, j]
                    B[8 * HALF + j, P] = frags[ks][j]
                acc[t] += A @ B
        return acc

    @staticmethod
    def frags_of(acc, relu):
        """accumulator tiles -> operand fragments of the next layer: register 16t + r = element r & 7 of k-step 2t + (r >> 3)."""
        regs = Wave.regs(acc)
        if relu:
            regs = np.maximum(regs, 0)
        return [regs[8 * ks:8 * ks + 8] for ks in range(regs.shape[0] // 8)], regs


def simulate_x3(idx, flat, emb, ins_num):
    T = (ins_num + 1 + 31) // 32
    vals = np.where(idx >= 0, flat[np.clip(idx, 0, None)], 0.0).astype(np.float64)
    cst, stream = vals[:CONST], vals[CONST:].copy()
    chunk_id = np.arange(stream.size) // X3_CHUNK
    stream[chunk_id % 2 == 1] = 0.0                               # lo chunks carry residuals: zero in exact arithmetic
    w = WaveX3(cst, stream)
    pe = np.zeros((32, 64)); qe = np.zeros((16, 64))
    for j in range(32):
        for h in range(2):
            s = pos_slot(j, h)
            if s >= 0:
                pe[j, HALF == h] = emb[:, s]
    for j in range(16):
        for h in range(2):
            s = pos_slot(j, h, n_freq_slots=12)
            if s >= 0:
                qe[j, HALF == h] = emb[:, 63 + s]
    pe_f = [pe[8 * k:8 * k + 8] for k in range(4)]
    qe_f = [qe[8 * k:8 * k + 8] for k in range(2)]
    act, _ = w.frags_of(w.layer(X3_STAGE["l0"], 8, pe_f, C_BIAS8), True)
    for l in range(1, 8):
        stage0 = X3_STAGE["l1"] + 4 * (l - 1) if l <= 4 else (X3_STAGE["l5"] if l == 5 else X3_STAGE["l6"] + 4 * (l - 6))
        act, regs = w.frags_of(w.layer(stage0, 8, act + (pe_f if l == 5 else []), C_BIAS8 + l * 256), True)
    sig = np.zeros(64)
    for i in range(128):
        sig += regs[i] * cst[C_DENSW + HALF * 128 + i]
    sigma = sig[:32] + sig[32:] + cst[C_DENSB]
    g, _ = w.frags_of(w.layer(X3_STAGE["rgbf"], 8, act, C_BIAS8 + 8 * 256), False)
    _, r = w.frags_of(w.layer(X3_STAGE["rgbh"], 4, g + qe_f, C_BIAS4), True)
    rgb = np.zeros((3, 32))
    for c in range(3):
        v = np.zeros(64)
        for i in range(64):
            v += r[i] * cst[C_RGBW + c * 128 + HALF * 64 + i]
        rgb[c] = v[:32] + v[32:] + cst[C_RGBB + c]
    e, _ = w.frags_of(w.layer(X3_STAGE["insf"], 8, act, C_BIAS8 + 9 * 256), False)
    s, _ = w.frags_of(w.layer(X3_STAGE["insh"], 4, e, C_BIAS4 + 128), True)
    # ins_linear bias slots are [h*64 + i] whatever T is
    acc3 = np.zeros((T, 32, 32))
    for t in range(T):
        for i in range(16):
            for h in range(2):
                acc3[t, (i & 3) + 8 * (i >> 2) + 4 * h, :] = cst[C_BIASO + h * 64 + 16 * t + i]
    nk = 8
    for t in range(T):
        for ks in range(nk):
            chunk = X3_STAGE["inso"] * 64 + (t * nk + ks) * 2
            a = stream[chunk * X3_CHUNK:(chunk + 1) * X3_CHUNK].reshape(64, 8)
            A = np.zeros((32, 16)); B = np.zeros((16, 32))
            for j in range(8):
                A[P, 8 * HALF + j] = a[:, j]
                B[8 * HALF + j, P] = s[ks][j]
            acc3[t] += A @ B
    ins = acc3.reshape(T * 32, 32)[: ins_num + 1]
    return np.concatenate([rgb, sigma[None], ins], 0).T


@pytest.mark.parametrize("ins_num,seed", [(13, 0), (93, 3)])
def test_x3_blob_consumed_in_mfma_order_equals_oracle_field(ins_num, seed):
    lib = dm_nerf_amd.load_library()
    n = lib.dmn_field_pack_index_x3_len(ins_num)
    assert n == CONST + 44 * 64 * X3_CHUNK
    assert lib.dmn_field_blob_x3_bytes(ins_num) == CONST * 4 + 44 * 64 * 1024
    idx = np.empty(n, dtype=np.int32)
    assert lib.dmn_field_pack_index_x3(ins_num, idx.ctypes.data) == 0
    params = oracle.make_params(ins_num, seed)
    flat = np.concatenate([params[k].numpy().reshape(-1) for k, _ in oracle.param_specs(ins_num)])
    x, v = cases.field_rows()
    emb = torch.cat([oracle.encode(torch.from_numpy(x[:32]), 10), oracle.encode(torch.from_numpy(v[:32]), 4)], -1)
    with torch.no_grad():
        want = oracle.field(params, emb).numpy()
    got = simulate_x3(idx, flat, emb.numpy().astype(np.float64), ins_num)
    assert rel_err(got, want) < 5e-6
    # every weight appears exactly twice in the stream (hi chunk and lo chunk), every bias/head value once
    used = np.bincount(idx[CONST:][idx[CONST:] >= 0], minlength=flat.size)
    assert set(np.unique(used)) <= {0, 2}
    both = np.bincount(idx[idx >= 0], minlength=flat.size)
    assert both.min() >= 1


# ---------------------------------------------------------------------------------------------
# full-precision bf16 ("x6") blob: three exact bf16 planes per weight, k-steps in the outer loop, 48-chunk stages
# (csrc/field_layout.h "x6", csrc/field_fwd_x6.hip).  The model below walks the stream the way the kernel does, splits
# the activations the way the kernel does (round to nearest even to bf16, exact residuals) and sums the kernel's six
# partial products -- in fp64, so that what is measured is the arithmetic's own error: the three dropped products.
# ---------------------------------------------------------------------------------------------
X6_STAGE = {"l0": 0, "l1": 2, "l5": 34, "l6": 44, "rgbf": 60, "rgbh": 68, "insf": 73, "insh": 81, "inso": 85}
X6_STAGE_CHUNKS = 48


def bf16_rne(x):
    """float32 array -> nearest-even bf16 value, as float32 (what v_cvt_pk_bf16_f32 and a cast to __bf16 give)."""
    b = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    b = (b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000
    return b.astype(np.uint32).view(np.float32)


def split3(x):
    x = np.asarray(x, dtype=np.float32)
    hi = bf16_rne(x)
    r1 = x - hi
    mid = bf16_rne(r1)
    lo = bf16_rne(r1 - mid)
    return hi, mid, lo


def test_three_way_bf16_split_is_exact():
    """hi + mid + lo == x bit for bit over the magnitudes the field sees (weights 1e-4 .. 30, activations to 7.7e3) and
    far beyond, and the residuals really are bf16 values: nothing of an fp32 operand is dropped by the x6 kernels'
    split.  (The one exception is the last binade below overflow: a value within 2^-9 of FLT_MAX rounds to a bf16
    infinity -- 3.39e38 is the largest value the split takes.)"""
    rng = np.random.default_rng(5)
    x = (rng.standard_normal(200000) * np.exp(rng.uniform(-12, 9, 200000))).astype(np.float32)
    x[:8] = [0.0, -0.0, 1.0, -1.0, 1.9999999, 255.99998, 1e-30, -3.38e38]
    hi, mid, lo = split3(x)
    assert np.array_equal((hi.astype(np.float64) + mid + lo).astype(np.float32), x)
    assert np.array_equal(hi.astype(np.float64) + mid.astype(np.float64) + lo.astype(np.float64), x.astype(np.float64))
    for plane in (hi, mid, lo):
        assert np.all(plane.view(np.uint32) & 0xFFFF == 0)
    # the six kept products against the exact one: each dropped term is below 2^-24 of |w x| (|mid| <= 2^-8 |x|,
    # |lo| <= 2^-16 |x|), all three together below 2^-23 and ~2^-26 in the root mean square -- the size of ONE fp32
    # rounding of the product, where an fp32 dot product of k terms carries k roundings of its partial sums
    w = (rng.standard_normal(200000) * np.exp(rng.uniform(-9, 3, 200000))).astype(np.float32)
    wh, wm, wl = (p.astype(np.float64) for p in split3(w))
    xh, xm, xl = (p.astype(np.float64) for p in (hi, mid, lo))
    six = wl * xh + wh * xl + wm * xm + wm * xh + wh * xm + wh * xh
    exact = w.astype(np.float64) * x.astype(np.float64)
    ok = exact != 0
    rel = np.abs(six - exact)[ok] / np.abs(exact)[ok]
    assert np.max(rel) < 2.0 ** -23 and np.sqrt(np.mean(rel ** 2)) < 2.0 ** -25.5, (np.max(rel), np.sqrt(np.mean(rel ** 2)))


class WaveX6:
    def __init__(self, cst, planes):
        self.cst, self.planes = cst, planes           # planes: the stream as (hi, mid, lo)-split values, one per element

    def layer(self, stage0, n_tiles, kps, frags, bias_off, bias_stride=None):
        """frags: per k-step [8][64] source values (element j, lane).  k-steps outer, tiles inner, six products."""
        acc = np.zeros((n_tiles, 32, 32))
        stride = 16 * n_tiles if bias_stride is None else bias_stride
        for t in range(n_tiles):
            for i in range(16):
                for h in range(2):
                    acc[t, (i & 3) + 8 * (i >> 2) + 4 * h, :] = self.cst[bias_off + h * stride + 16 * t + i]
        for ks, f in enumerate(frags):
            bh, bm, bl = (p.astype(np.float64) for p in split3(f))
            for t in range(n_tiles):
                chunk = (stage0 + ks // kps) * X6_STAGE_CHUNKS + ((ks % kps) * n_tiles + t) * 3
                A = []
                for plane in range(3):
                    a = self.planes[(chunk + plane) * X3_CHUNK:(chunk + plane + 1) * X3_CHUNK].reshape(64, 8)
                    M = np.zeros((32, 16))
                    for j in range(8):
                        M[P, 8 * HALF + j] = a[:, j]
                    A.append(M)
                B = []
                for b in (bh, bm, bl):
                    M = np.zeros((16, 32))
                    for j in range(8):
                        M[8 * HALF + j, P] = b[j]
                    B.append(M)
                ah, am, al = A
                Bh, Bm, Bl = B
                acc[t] += al @ Bh + ah @ Bl + am @ Bm + am @ Bh + ah @ Bm + ah @ Bh
        return acc


def simulate_x6(idx, flat, emb, ins_num):
    T = (ins_num + 1 + 31) // 32
    vals = np.where(idx >= 0, flat[np.clip(idx, 0, None)], 0.0).astype(np.float32)
    cst = vals[:CONST].astype(np.float64)
    stream = vals[CONST:]
    hi, mid, lo = split3(stream)
    plane = (np.arange(stream.size) // X3_CHUNK) % 3              # chunk position modulo 3 (pack_x6_kernel)
    planes = np.where(plane == 0, hi, np.where(plane == 1, mid, lo)).astype(np.float64)
    w = WaveX6(cst, planes)
    pe = np.zeros((32, 64), dtype=np.float32); qe = np.zeros((16, 64), dtype=np.float32)
    for j in range(32):
        for h in range(2):
            s = pos_slot(j, h)
            if s >= 0:
                pe[j, HALF == h] = emb[:, s]
    for j in range(16):
        for h in range(2):
            s = pos_slot(j, h, n_freq_slots=12)
            if s >= 0:
                qe[j, HALF == h] = emb[:, 63 + s]
    pe_f = [pe[8 * k:8 * k + 8] for k in range(4)]
    qe_f = [qe[8 * k:8 * k + 8] for k in range(2)]

    def frags_of(acc, relu):
        regs = Wave.regs(acc).astype(np.float32)                  # the accumulators are fp32 registers
        if relu:
            regs = np.maximum(regs, 0)
        return [regs[8 * ks:8 * ks + 8] for ks in range(regs.shape[0] // 8)], regs

    act, _ = frags_of(w.layer(X6_STAGE["l0"], 8, 2, pe_f, C_BIAS8), True)
    for l in range(1, 8):
        stage0 = X6_STAGE["l1"] + 8 * (l - 1) if l <= 4 else (X6_STAGE["l5"] if l == 5 else X6_STAGE["l6"] + 8 * (l - 6))
        act, regs = frags_of(w.layer(stage0, 8, 2, act + (pe_f if l == 5 else []), C_BIAS8 + l * 256), True)
    sig = np.zeros(64)
    for i in range(128):
        sig += regs[i].astype(np.float64) * cst[C_DENSW + HALF * 128 + i]
    sigma = sig[:32] + sig[32:] + cst[C_DENSB]
    g, _ = frags_of(w.layer(X6_STAGE["rgbf"], 8, 2, act, C_BIAS8 + 8 * 256), False)
    _, r = frags_of(w.layer(X6_STAGE["rgbh"], 4, 4, g + qe_f, C_BIAS4), True)
    rgb = np.zeros((3, 32))
    for c in range(3):
        v = np.zeros(64)
        for i in range(64):
            v += r[i].astype(np.float64) * cst[C_RGBW + c * 128 + HALF * 64 + i]
        rgb[c] = v[:32] + v[32:] + cst[C_RGBB + c]
    e, _ = frags_of(w.layer(X6_STAGE["insf"], 8, 2, act, C_BIAS8 + 9 * 256), False)
    s, _ = frags_of(w.layer(X6_STAGE["insh"], 4, 4, e, C_BIAS4 + 128), True)
    acc3 = w.layer(X6_STAGE["inso"], T, 8 if T <= 2 else 4, s, C_BIASO, bias_stride=64)
    ins = acc3.reshape(T * 32, 32)[: ins_num + 1]
    return np.concatenate([rgb, sigma[None], ins], 0).T


@pytest.mark.parametrize("ins_num,seed", [(13, 0), (93, 3), (40, 5)])
def test_x6_blob_consumed_in_kernel_order_equals_the_field_to_fp32_rounding(ins_num, seed):
    lib = dm_nerf_amd.load_library()
    T = (ins_num + 1 + 31) // 32
    stages = 85 + (1 if T <= 2 else 2)
    n = lib.dmn_field_pack_index_x6_len(ins_num)
    assert n == CONST + stages * X6_STAGE_CHUNKS * X3_CHUNK
    assert lib.dmn_field_blob_x6_bytes(ins_num) == CONST * 4 + stages * X6_STAGE_CHUNKS * 1024
    idx = np.empty(n, dtype=np.int32)
    assert lib.dmn_field_pack_index_x6(ins_num, idx.ctypes.data) == 0
    params = oracle.make_params(ins_num, seed)
    flat = np.concatenate([params[k].numpy().reshape(-1) for k, _ in oracle.param_specs(ins_num)])
    x, v = cases.field_rows()
    emb = torch.cat([oracle.encode(torch.from_numpy(x[:32]), 10), oracle.encode(torch.from_numpy(v[:32]), 4)], -1)
    with torch.no_grad():
        want32 = oracle.field(params, emb).numpy()
        want64 = oracle.field({k: p.double() for k, p in params.items()}, emb.double()).numpy()
    got = simulate_x6(idx, flat, emb.numpy(), ins_num)
    assert got.shape == want64.shape
    # against the fp64 evaluation of the same fp32 parameters: the six-product arithmetic with fp32 layer outputs is
    # as close to it as the reference's own fp32 evaluation is (both are bounded by the rounding of the activations)
    err_x6, err_ref = rel_err(got, want64), rel_err(want32, want64)
    assert err_x6 < 2e-6 and err_x6 < 2.0 * err_ref + 1e-7, (err_x6, err_ref)
    # every weight appears exactly three times in the stream (hi, mid and lo chunk), every bias / head value once
    used = np.bincount(idx[CONST:][idx[CONST:] >= 0], minlength=flat.size)
    assert set(np.unique(used)) <= {0, 3}
    assert np.bincount(idx[idx >= 0], minlength=flat.size).min() >= 1


# ---------------------------------------------------------------------------------------------
# "x6w" blob: the same planes for v_mfma_f32_16x16x32_bf16 (csrc/field_layout.h "x6w", csrc/field_fwd_x6w.hip): 16 x 16
# tiles, 32-deep k-steps, two point tiles per wave, const section in natural feature order.  Lane-exact model of the
# instruction's operand / accumulator maps: A[i = l & 15][k = 8 (l >> 4) + j], B[k = 8 (l >> 4) + j][n = l & 15],
# D[i = 4 (l >> 4) + r][n = l & 15].
# ---------------------------------------------------------------------------------------------
N16, Q16 = LANE & 15, LANE >> 4


class WaveX6W:
    def __init__(self, cst, planes):
        self.cst, self.planes = cst, planes

    def layer(self, stage0, n_tiles, slots, bias_off):
        """slots: per k-step [2 point tiles][8 k-slots][64 lanes] source values.  Returns acc[p][t][16 features][16 points]."""
        kps = 16 // n_tiles
        acc = np.zeros((2, n_tiles, 16, 16))
        for t in range(n_tiles):
            acc[:, t] = self.cst[bias_off + 16 * t:bias_off + 16 * t + 16][None, :, None]
        for ks, f in enumerate(slots):
            for t in range(n_tiles):
                chunk = (stage0 + ks // kps) * X6_STAGE_CHUNKS + ((ks % kps) * n_tiles + t) * 3
                A = []
                for plane in range(3):
                    a = self.planes[(chunk + plane) * X3_CHUNK:(chunk + plane + 1) * X3_CHUNK].reshape(64, 8)
                    M = np.zeros((16, 32))
                    for j in range(8):
                        M[N16, 8 * Q16 + j] = a[:, j]
                    A.append(M)
                ah, am, al = A
                for p in range(2):
                    B = []
                    for b in split3(f[p]):
                        M = np.zeros((32, 16))
                        for j in range(8):
                            M[8 * Q16 + j, N16] = b[j].astype(np.float64)
                        B.append(M)
                    Bh, Bm, Bl = B
                    acc[p, t] += al @ Bh + ah @ Bl + am @ Bm + am @ Bh + ah @ Bm + ah @ Bh
        return acc

    @staticmethod
    def slots_of(acc, relu):
        """k-slots the NEXT layer reads out of these accumulators: k-step ks, slot j of lane (n, q) is register j & 3 of
        feature tile 2 ks + (j >> 2), i.e. feature 16 (2 ks + (j >> 2)) + 4 q + (j & 3) of point n."""
        vals = acc.astype(np.float32)
        if relu:
            vals = np.maximum(vals, 0)
        n_tiles = acc.shape[1]
        out = []
        for ks in range(n_tiles // 2):
            f = np.zeros((2, 8, 64), dtype=np.float32)
            for p in range(2):
                for j in range(8):
                    f[p, j] = vals[p, 2 * ks + (j >> 2), 4 * Q16 + (j & 3), N16]
            out.append(f)
        return out, vals


def simulate_x6w(idx, flat, emb, ins_num):
    T16 = (ins_num + 1 + 15) // 16
    T16 = T16 if T16 <= 2 else (4 if T16 <= 4 else (6 if T16 <= 6 else 8))
    vals = np.where(idx >= 0, flat[np.clip(idx, 0, None)], 0.0).astype(np.float32)
    cst = vals[:CONST].astype(np.float64)
    stream = vals[CONST:]
    hi, mid, lo = split3(stream)
    plane = (np.arange(stream.size) // X3_CHUNK) % 3
    w = WaveX6W(cst, np.where(plane == 0, hi, np.where(plane == 1, mid, lo)).astype(np.float64))

    def enc_slots(ch0, channels, n_ksteps):          # channel ch0 + 32 e + 8 q + j of point 16 p + n, zero past the end
        out = []
        for e in range(n_ksteps):
            f = np.zeros((2, 8, 64), dtype=np.float32)
            for p in range(2):
                for j in range(8):
                    c = 32 * e + 8 * Q16 + j
                    ok = c < channels
                    f[p, j, ok] = emb[16 * p + N16[ok], ch0 + c[ok]]
            out.append(f)
        return out

    pe, qe = enc_slots(0, 63, 2), enc_slots(63, 27, 1)
    act, _ = WaveX6W.slots_of(w.layer(X6_STAGE["l0"], 16, pe, C_BIAS8), True)
    for l in range(1, 8):
        stage0 = X6_STAGE["l1"] + 8 * (l - 1) if l <= 4 else (X6_STAGE["l5"] if l == 5 else X6_STAGE["l6"] + 8 * (l - 6))
        act, h = WaveX6W.slots_of(w.layer(stage0, 16, act + (pe if l == 5 else []), C_BIAS8 + l * 256), True)
    # heads in natural feature order: h[p][t][f][n] -> feature 16 t + f
    h8 = h.reshape(2, 256, 16).astype(np.float64)
    sigma = np.einsum("pfn,f->pn", h8, cst[C_DENSW:C_DENSW + 256]) + cst[C_DENSB]
    g, _ = WaveX6W.slots_of(w.layer(X6_STAGE["rgbf"], 16, act, C_BIAS8 + 8 * 256), False)
    _, r = WaveX6W.slots_of(w.layer(X6_STAGE["rgbh"], 8, g + qe, C_BIAS4), True)
    r = r.reshape(2, 128, 16).astype(np.float64)
    rgb = np.stack([np.einsum("pfn,f->pn", r, cst[C_RGBW + c * 128:C_RGBW + c * 128 + 128]) + cst[C_RGBB + c] for c in range(3)], 1)
    e, _ = WaveX6W.slots_of(w.layer(X6_STAGE["insf"], 16, act, C_BIAS8 + 9 * 256), False)
    s, _ = WaveX6W.slots_of(w.layer(X6_STAGE["insh"], 8, e, C_BIAS4 + 128), True)
    acc3 = w.layer(X6_STAGE["inso"], T16, s, C_BIASO)
    ins = acc3.reshape(2, T16 * 16, 16)[:, : ins_num + 1]
    out = np.concatenate([rgb, sigma[:, None], ins], 1)            # [p][channel][n]
    return np.concatenate([out[0].T, out[1].T], 0)                 # [32 points][channels]


@pytest.mark.parametrize("ins_num,seed", [(13, 0), (40, 5), (93, 3), (127, 4)])
def test_x6w_blob_consumed_in_kernel_order_equals_the_field_to_fp32_rounding(ins_num, seed):
    lib = dm_nerf_amd.load_library()
    n = lib.dmn_field_pack_index_x6_len(ins_num)               # same size as the x6 blob: same bytes, other order
    idx = np.empty(n, dtype=np.int32)
    assert lib.dmn_field_pack_index_x6w(ins_num, idx.ctypes.data) == 0
    params = oracle.make_params(ins_num, seed)
    flat = np.concatenate([params[k].numpy().reshape(-1) for k, _ in oracle.param_specs(ins_num)])
    x, v = cases.field_rows()
    emb = torch.cat([oracle.encode(torch.from_numpy(x[:32]), 10), oracle.encode(torch.from_numpy(v[:32]), 4)], -1)
    with torch.no_grad():
        want32 = oracle.field(params, emb).numpy()
        want64 = oracle.field({k: p.double() for k, p in params.items()}, emb.double()).numpy()
    got = simulate_x6w(idx, flat, emb.numpy(), ins_num)
    assert got.shape == want64.shape
    err_w, err_ref = rel_err(got, want64), rel_err(want32, want64)
    assert err_w < 2e-6 and err_w < 2.0 * err_ref + 1e-7, (err_w, err_ref)
    # the same multiset of parameters as the x6 blob: every weight three times (hi, mid, lo chunk), biases / heads once
    idx6 = np.empty(n, dtype=np.int32)
    assert lib.dmn_field_pack_index_x6(ins_num, idx6.ctypes.data) == 0
    np.testing.assert_array_equal(np.bincount(idx[idx >= 0], minlength=flat.size), np.bincount(idx6[idx6 >= 0], minlength=flat.size))


def test_six_product_dot_products_are_as_accurate_as_an_fp32_fma_chain():
    """A 256-term dot product (ReLU-like activations, weights of a 256-wide layer) against its exact value: the six-product
    arithmetic with fp32 accumulation per 16-term block (what a bf16 MFMA does) is no further away than the k-ordered
    fp32 fma chain of the exact-fp32 MFMA, and adding the three dropped products (all nine) changes nothing -- the error of
    both is the rounding of the fp32 accumulator, of which the chain has 256 per dot product and the block form 96.  The
    two-way bf16 split (three products) is an order of magnitude further out."""
    rng = np.random.default_rng(0)
    K, N = 256, 4000
    w = (rng.uniform(-1, 1, (N, K)) / 16).astype(np.float32)
    x = np.maximum(rng.standard_normal((N, K)), 0).astype(np.float32)
    w64, x64 = w.astype(np.float64), x.astype(np.float64)
    exact = (w64 * x64).sum(1)
    acc = np.zeros(N, dtype=np.float32)
    for k in range(K):                                            # v_mfma_f32_32x32x2_f32: a k-ordered fma chain
        acc = (acc.astype(np.float64) + w64[:, k] * x64[:, k]).astype(np.float32)
    chain = acc
    wh, wm, wl = (p.astype(np.float64) for p in split3(w))
    xh, xm, xl = (p.astype(np.float64) for p in split3(x))

    def blocks(pairs):
        acc = np.zeros(N, dtype=np.float32)
        for b in range(0, K, 16):
            sl = slice(b, b + 16)
            for A, B in pairs:
                acc = (acc.astype(np.float64) + (A[:, sl] * B[:, sl]).sum(1)).astype(np.float32)
        return acc

    six = blocks(((wl, xh), (wh, xl), (wm, xm), (wm, xh), (wh, xm), (wh, xh)))
    nine = blocks(((wl, xl), (wl, xm), (wm, xl), (wl, xh), (wh, xl), (wm, xm), (wm, xh), (wh, xm), (wh, xh)))
    wh2, xh2 = bf16_rne(w), bf16_rne(x)
    wl2, xl2 = bf16_rne(w - wh2), bf16_rne(x - xh2)
    three = blocks(tuple((a.astype(np.float64), b.astype(np.float64)) for a, b in ((wh2, xl2), (wl2, xh2), (wh2, xh2))))
    rms = lambda v: float(np.sqrt(np.mean((v.astype(np.float64) - exact) ** 2)) / np.abs(exact).max())
    assert rms(six) <= 1.05 * rms(chain), (rms(six), rms(chain))
    assert abs(rms(nine) - rms(six)) < 0.02 * rms(six)
    assert rms(three) > 10 * rms(six)
