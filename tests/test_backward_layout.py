"""CPU checks of the host tables behind the field backward (csrc/field_layout.h "Training", field_pack.cpp):
the weight-gradient scatter table and the transposed-weight blobs of the backward-data kernels.  The kernels
themselves are checked on the GPU (tests/test_gpu_parity.py)."""
import numpy as np
import pytest

from oracle import dmnerf_cpu as oracle

import dm_nerf_amd
from test_layout import CONST, CHUNK, STAGE, X3_CHUNK, HALF, P, feat_of, pos_slot

# save / gradient record register offsets (field_layout.h)
SAVE_PE, SAVE_QE, SAVE_H, SAVE_FEAT, SAVE_R1, SAVE_INSF, SAVE_I1 = 0, 32, 48, 1072, 1200, 1264, 1392
GRAD_A, GRAD_FEAT, GRAD_R1, GRAD_INSF, GRAD_I1, GRAD_HEAD = 0, 1024, 1152, 1216, 1344, 1408


def _offsets(ins_num):
    off, at = {}, 0
    for name, shape in oracle.param_specs(ins_num):
        off[name] = (at, shape)
        at += int(np.prod(shape))
    return off, at


def _table(ins_num):
    lib = dm_nerf_amd.load_library()
    n = lib.dmn_field_wgrad_index_len(ins_num)
    idx = np.empty(n, dtype=np.int32)
    assert lib.dmn_field_wgrad_index(ins_num, idx.ctypes.data) == 0
    return idx


@pytest.mark.parametrize("ins_num", [13, 40, 93])
def test_weight_gradient_table_covers_every_parameter_exactly_once(ins_num):
    lib = dm_nerf_amd.load_library()
    idx = _table(ins_num)
    total = lib.dmn_field_param_floats(ins_num)
    used = idx[idx >= 0]
    counts = np.bincount(used, minlength=total)
    assert counts.min() == 1 and counts.max() == 1 and used.max() == total - 1
    T = (ins_num + 1 + 31) // 32
    # element space: nine 256x256 GEMMs + their biases first, then the small ones (field_layout.h kWgradBigLen)
    small = (2 * 256 * 64 + 2 * 128 * 256 + 128 * 32 + 32 * T * 128 + 32 * 128 + 32 * 256) + (256 + 128 + 128 + 32 * T + 32 + 32)
    assert idx.size == 9 * (256 * 256 + 256) + small
    assert lib.dmn_field_wgrad_workspace_floats(ins_num, 10) == 10 * (9 * (256 * 256 + 256) + 3 * small)


def _image_rows(x, kind):
    """x: [features, points] -> image rows [2*regs, points] (row r = 2*i + h <-> feat_of / pos_slot / dir_slot)."""
    regs = {"act256": 128, "act128": 64, "pe": 32, "qe": 16}[kind]
    out = np.zeros((2 * regs, x.shape[1]))
    for i in range(regs):
        for h in range(2):
            if kind == "pe":
                f = pos_slot(i, h)
            elif kind == "qe":
                f = pos_slot(i, h, n_freq_slots=12)
            else:
                f = feat_of(i, h)
            if f >= 0:
                out[2 * i + h] = x[f]
    return out


def test_weight_gradient_table_scatters_row_products_onto_the_right_parameters():
    """Emulate the weight-gradient kernel in numpy (G rows x X rows over the points, images in register order) and
    scatter through the table: must equal autograd of the same linear layers."""
    ins_num, n = 13, 24
    T = 1
    rng = np.random.default_rng(7)
    params = oracle.make_params(ins_num, 2)
    off, total = _offsets(ins_num)
    idx = _table(ins_num)

    # random layer inputs X and output gradients G in feature space, [features, points]
    pe, qe = rng.standard_normal((63, n)), rng.standard_normal((27, n))
    h = {l: rng.standard_normal((256, n)) for l in range(1, 9)}
    feat, insf = rng.standard_normal((256, n)), rng.standard_normal((256, n))
    r1, i1 = rng.standard_normal((128, n)), rng.standard_normal((128, n))
    G = {l: rng.standard_normal((256, n)) for l in range(8)}
    g_feat, g_insf = rng.standard_normal((256, n)), rng.standard_normal((256, n))
    g_r1, g_i1 = rng.standard_normal((128, n)), rng.standard_normal((128, n))
    g_raw = rng.standard_normal((5 + ins_num, n))                     # rgb(3), sigma, ins(ins_num + 1)

    want = np.zeros(total)

    def put(name, value):
        at, shape = off[name]
        want[at:at + value.size] += value.reshape(-1)

    put("mlps.0.weight", G[0] @ pe.T); put("mlps.0.bias", G[0].sum(1))
    for l in range(1, 8):
        x = np.concatenate([h[l], pe], 0) if l == 5 else h[l]
        put(f"mlps.{l}.weight", G[l] @ x.T); put(f"mlps.{l}.bias", G[l].sum(1))
    put("rgb_feature_linear.weight", g_feat @ h[8].T); put("rgb_feature_linear.bias", g_feat.sum(1))
    put("ins_feature_linear.weight", g_insf @ h[8].T); put("ins_feature_linear.bias", g_insf.sum(1))
    put("rgb_feature_linears.0.weight", g_r1 @ np.concatenate([feat, qe], 0).T); put("rgb_feature_linears.0.bias", g_r1.sum(1))
    put("ins_feature_linears.0.weight", g_i1 @ insf.T); put("ins_feature_linears.0.bias", g_i1.sum(1))
    put("density_linear.weight", g_raw[3:4] @ h[8].T); put("density_linear.bias", g_raw[3:4].sum(1))
    put("ins_linear.weight", g_raw[4:] @ i1.T); put("ins_linear.bias", g_raw[4:].sum(1))
    put("rgb_linear.weight", g_raw[:3] @ r1.T); put("rgb_linear.bias", g_raw[:3].sum(1))

    # images: save record rows and gradient record rows, keyed by register offset
    save = {SAVE_PE: _image_rows(pe, "pe"), SAVE_QE: _image_rows(qe, "qe"), SAVE_FEAT: _image_rows(feat, "act256"),
            SAVE_R1: _image_rows(r1, "act128"), SAVE_INSF: _image_rows(insf, "act256"), SAVE_I1: _image_rows(i1, "act128")}
    for l in range(1, 9):
        save[SAVE_H + 128 * (l - 1)] = _image_rows(h[l], "act256")
    grad = {GRAD_FEAT: _image_rows(g_feat, "act256"), GRAD_R1: _image_rows(g_r1, "act128"),
            GRAD_INSF: _image_rows(g_insf, "act256"), GRAD_I1: _image_rows(g_i1, "act128")}
    for l in range(8):
        grad[GRAD_A + 128 * l] = _image_rows(G[l], "act256")
    head_ins = np.zeros((32 * T, n))
    for i in range(16 * T):
        for hh in range(2):
            ch = feat_of(i, hh)
            if ch <= ins_num:
                head_ins[2 * i + hh] = g_raw[4 + ch]
    head_rgb = np.zeros((32, n))
    head_rgb[0], head_rgb[1], head_rgb[2], head_rgb[3] = g_raw[0], g_raw[1], g_raw[2], g_raw[3]
    grad[GRAD_HEAD], grad[GRAD_HEAD + 16 * T] = head_ins, head_rgb

    # the GEMM list of field_layout.h wgemm_table: (gradient register, save register, bias?)
    gemms = [(GRAD_A, SAVE_PE, True)] + [(GRAD_A + 128 * l, SAVE_H + 128 * (l - 1), True) for l in range(1, 5)]
    gemms += [(GRAD_A + 640, SAVE_H + 512, True), (GRAD_A + 640, SAVE_PE, False), (GRAD_A + 768, SAVE_H + 640, True),
              (GRAD_A + 896, SAVE_H + 768, True), (GRAD_FEAT, SAVE_H + 896, True), (GRAD_INSF, SAVE_H + 896, True),
              (GRAD_R1, SAVE_FEAT, True), (GRAD_R1, SAVE_QE, False), (GRAD_I1, SAVE_INSF, True),
              (GRAD_HEAD, SAVE_I1, True), (GRAD_HEAD + 16 * T, SAVE_R1, True), (GRAD_HEAD + 16 * T, SAVE_H + 896, True)]
    big = [g for g in gemms if grad[g[0]].shape[0] == 256 and save[g[1]].shape[0] == 256]
    order = big + [g for g in gemms if g not in big]
    got = np.zeros(total)
    at = 0
    blocks = []
    for pass_gemms in (big, [g for g in gemms if g not in big]):
        for g_reg, x_reg, _ in pass_gemms:                              # tables of the pass, then its biases
            blocks.append((grad[g_reg] @ save[x_reg].T).reshape(-1))
        for g_reg, x_reg, bias in pass_gemms:
            if bias:
                blocks.append(grad[g_reg].sum(1))
    flat = np.concatenate(blocks)
    assert flat.size == idx.size and len(order) == 17
    ok = idx >= 0
    np.add.at(got, idx[ok], flat[ok])
    np.testing.assert_allclose(got, want, rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("ins_num", [13, 93])
def test_backward_data_blobs_hold_the_transposed_weights(ins_num):
    """fp32 and x3 backward-data tables: every weight the chain multiplies by appears once (x3: as a hi and a lo
    chunk), nothing else appears, and chunk element [lane][e] of a k-step is W[k feature][row] (transposed)."""
    lib = dm_nerf_amd.load_library()
    off, total = _offsets(ins_num)
    T = (ins_num + 1 + 31) // 32
    n = lib.dmn_field_bwd_blob_floats(ins_num)
    assert n == CONST + (72 + (1 if T <= 2 else 2)) * STAGE
    idx = np.empty(n, dtype=np.int32)
    assert lib.dmn_field_bwd_pack_index(ins_num, idx.ctypes.data) == 0
    nx = lib.dmn_field_bwd_pack_index_x3_len(ins_num)
    assert nx == CONST + 37 * 64 * X3_CHUNK and lib.dmn_field_bwd_blob_x3_bytes(ins_num) == CONST * 4 + 37 * 64 * 1024
    idx3 = np.empty(nx, dtype=np.int32)
    assert lib.dmn_field_bwd_pack_index_x3(ins_num, idx3.ctypes.data) == 0

    expect = np.zeros(total, dtype=np.int64)                        # how often the stream holds each parameter

    def mark(name, rows=None, cols=None):
        at, shape = off[name]
        m = np.zeros(shape, dtype=np.int64)
        m[slice(None) if rows is None else rows, slice(None) if cols is None else cols] = 1
        expect[at:at + m.size] += m.reshape(-1)

    for l in range(1, 8):
        mark(f"mlps.{l}.weight", cols=slice(0, 256))                 # d/d h only: the skip's input_pts needs no gradient
    mark("rgb_feature_linear.weight")
    mark("rgb_feature_linears.0.weight", cols=slice(0, 256))        # feature columns, not the view encoding
    mark("ins_feature_linears.0.weight")
    mark("ins_linear.weight")                                        # ins_feature_linear: the branch is detached there
    stream = idx[CONST:]
    np.testing.assert_array_equal(np.bincount(stream[stream >= 0], minlength=total), expect)
    stream3 = idx3[CONST:]
    np.testing.assert_array_equal(np.bincount(stream3[stream3 >= 0], minlength=total), 2 * expect)
    np.testing.assert_array_equal(idx[:CONST], idx3[:CONST])        # same const section as the forward blobs
    assert (idx[3592:CONST] == -1).all()                            # zero padding = initial accumulators of the x3 kernel

    # spot check of the transposition, fp32 stream: mlps.3^T is the 5th 8-stage block from the end (l = 7..1)
    s0 = (1 if T <= 2 else 2) + 4 + 4 + 8 + 8 * (7 - 3)
    at, shape = off["mlps.3.weight"]
    for j in (0, 17, 127):
        for q in range(2):
            c = stream[(s0 * 32 + 2 * j + q) * CHUNK:(s0 * 32 + 2 * j + q + 1) * CHUNK].reshape(64, 4)
            for e in range(4):
                rows = 32 * (4 * q + e) + P
                kf = feat_of(j, HALF)
                np.testing.assert_array_equal(c[:, e], at + kf * shape[1] + rows)
    # x3 stream: tile t, k-step ks, element jj: W[feat_of(8 ks + jj, h)][32 t + lane & 31], hi and lo chunk alike
    s0 = 5 + 4 + 4 * (7 - 3)
    for t, ks in ((0, 0), (5, 9), (7, 15)):
        for part in range(2):
            ch = s0 * 64 + (t * 16 + ks) * 2 + part
            c = stream3[ch * X3_CHUNK:(ch + 1) * X3_CHUNK].reshape(64, 8)
            for jj in range(8):
                np.testing.assert_array_equal(c[:, jj], at + feat_of(8 * ks + jj, HALF) * shape[1] + 32 * t + P)

    # x6 stream (field_bwd_x6.hip): k-steps outer, [hi, mid, lo] chunk triples per (k-step, tile) group, 48-chunk stages;
    # every weight the chain multiplies by three times, nothing else; group (ks, t) of mlps.3^T, all three planes alike
    n6 = lib.dmn_field_bwd_pack_index_x6_len(ins_num)
    stages6 = (1 if T <= 2 else 2) + 4 + 4 + 8 + 7 * 8
    assert n6 == CONST + stages6 * 48 * X3_CHUNK and lib.dmn_field_bwd_blob_x6_bytes(ins_num) == CONST * 4 + stages6 * 48 * 1024
    idx6 = np.empty(n6, dtype=np.int32)
    assert lib.dmn_field_bwd_pack_index_x6(ins_num, idx6.ctypes.data) == 0
    stream6 = idx6[CONST:]
    np.testing.assert_array_equal(np.bincount(stream6[stream6 >= 0], minlength=total), 3 * expect)
    np.testing.assert_array_equal(idx6[:CONST], idx[:CONST])
    s0 = (1 if T <= 2 else 2) + 4 + 4 + 8 + 8 * (7 - 3)
    for t, ks in ((0, 0), (5, 9), (7, 15)):
        for plane in range(3):
            ch = (s0 + ks // 2) * 48 + ((ks % 2) * 8 + t) * 3 + plane
            c = stream6[ch * X3_CHUNK:(ch + 1) * X3_CHUNK].reshape(64, 8)
            for jj in range(8):
                np.testing.assert_array_equal(c[:, jj], at + feat_of(8 * ks + jj, HALF) * shape[1] + 32 * t + P)
    # ins_linear^T heads the stream: 4 tiles x 2 T k-steps, 4 k-steps per stage; channels past ins_num are padding
    at_o, shape_o = off["ins_linear.weight"]
    c = stream6[0:X3_CHUNK].reshape(64, 8)                                   # k-step 0, tile 0, hi plane
    for jj in range(8):
        ch_k = feat_of(jj, HALF)                                             # channel held by k-slot jj of lane-half h
        want = np.where(ch_k <= ins_num, at_o + ch_k * shape_o[1] + P, -1)
        np.testing.assert_array_equal(c[:, jj], want)
