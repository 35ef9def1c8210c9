"""CPU check of the packed-weight layout (csrc/field_layout.h + field_pack.cpp).

The blob built from the library's gather table is consumed here by a lane-exact model of
v_mfma_f32_32x32x2_f32 (operand and accumulator lane maps as documented for gfx950) that walks
the weight stream exactly the way the fused kernel does: activations as the B operand straight
from the accumulator registers, weights as the A operand from 1 KiB chunks.  The result must equal
the oracle's field network.  This pins the layout logic on CPU; the kernel itself is checked on
the GPU (tests/test_gpu_parity.py)."""
import ctypes

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
