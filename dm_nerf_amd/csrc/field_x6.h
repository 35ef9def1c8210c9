// Full-precision bf16 ("bf16x6") machinery shared by the field kernel (field_fwd_x6.hip) and the fused field + compositing
// kernel (field_fused_x3.hip): the 48 KiB-stage weight stream with its spread-out LDS-DMA, the exact three-way operand
// split, the k-steps-outer layer loop on two alternating accumulator sets, and the whole network as one device function.
//
// The arithmetic: the same fused network as field_fwd.hip (encodings + 8x256 trunk with skip + density / colour /
// instance heads, networks/render.py:37-38,49-61,71-83 and networks/dm_nerf.py:80-106), with every fp32 product W*X
// evaluated from EXACT three-way bf16 splits
//     w = wh + wm + wl,   x = xh + xm + xl        (8 + 8 + 8 significand bits: nothing of the fp32 value is lost)
//     W*X ~= Wl*Xh + Wh*Xl + Wm*Xm + Wm*Xh + Wh*Xm + Wh*Xh
// on v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  The three dropped terms (Wm*Xl, Wl*Xm, Wl*Xl) are each below
// 2^-24 of the product (together 2^-23 at worst, ~2^-26 rms: tests/test_layout.py) -- the size of ONE fp32 rounding of
// the product, against the k roundings of partial sums an fp32 dot product of k terms carries -- so the arithmetic
// has fp32's operand precision, unlike the two-way "x3" kernels (16 operand bits).  Six 32-cycle MFMAs cover 16
// k-values where the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32, 64 cycles per 2 k-values) needs eight: 2.7x its matrix
// throughput.
//
// Structure = the exact kernel's (field_fwd.hip), not the x3 kernel's: k-steps in the OUTER loop, all output tiles of
// a layer accumulate at once, so a k-step's operand fragments are split once (about 15 vector instructions per value
// pair) and used by 6 x 8 MFMAs.  Activations stay fp32 in the accumulator registers of the layer that produced them:
// two accumulator sets (2 x 128 registers) alternate as input and output, a k-step reads its 8 values per lane out
// of the finished set, applies ReLU, splits them into three packed fragments -- one k-step ahead, in the shadow of the
// running k-step's MFMAs.  Weights: three 1 KiB chunks [hi, mid, lo] per (k-step, tile), 48 KiB stages, LDS-DMA double
// buffered, one piece per (k-step, tile) group behind that group's MFMAs (field_layout.h "x6").
#pragma once
#include "field_common.h"
#include "field_x3.h"

namespace dmn {

typedef PrecBF16 P6;
typedef P6::frag frag;

// Stage slots in LDS.  2: the DMA of stage g+1 runs during stage g and is waited for at the barrier that begins stage
// g+1.  3 (159 KiB of the CU's 160): the DMA of stage g+2 runs during stage g; DMN_X6_XBAR = 0 waits at the barrier of
// stage g only for the pieces issued a whole stage earlier (counted vmcnt: the 12 pieces of stage g+1 stay in flight),
// DMN_X6_XBAR = 1 waits for everything there, so that stage g+1 is complete one barrier early and its first fragments
// can be read ahead of its own barrier.
#ifndef DMN_X6_SLOTS
#define DMN_X6_SLOTS 2
#endif
#ifndef DMN_X6_XBAR
#define DMN_X6_XBAR 0
#endif
#ifndef DMN_X6_ORDER
#define DMN_X6_ORDER 0
#endif
#ifndef DMN_X6_ABL
#define DMN_X6_ABL 0
#endif
#ifndef DMN_X6_SCALAR_SUB
#define DMN_X6_SCALAR_SUB 1
#endif
#ifndef DMN_X6_NINE
#define DMN_X6_NINE 0
#endif
// Layers (bit = layer id below) whose LDS-DMA pieces are issued with the non-temporal cache policy: the three-plane
// stream of a network (4.03 MiB) is a cyclic working set just larger than the XCD's 4 MiB L2, which under LRU misses on
// EVERY line; marking a part of it for early replacement leaves a remainder that fits and stays (experiments/README.md,
// Round 4).  Ids: 0 mlps.0, 1 mlps.1-4, 2 mlps.5, 3 mlps.6, 4 mlps.7, 5 rgb_feature_linear, 6 rgb_feature_linears.0,
// 7 ins_feature_linear, 8 ins_feature_linears.0, 9 ins_linear.  (The pieces a layer issues belong to the stage AFTER
// the running one, so the marked range is shifted by one stage against the layer: irrelevant for the purpose.)
#ifndef DMN_X6_NTMASK
#define DMN_X6_NTMASK 0x380
#endif
constexpr int kX6Slots = DMN_X6_SLOTS;
static_assert(kX6Slots == 2 || kX6Slots == 3, "two or three stage slots");
constexpr bool kX6CrossBarrier = kX6Slots == 3 && DMN_X6_XBAR;
constexpr int kX6LdsBytes = kX6Slots * kX6StageBytes + kConstFloats * 4;      // 113,664 B (2 slots) / 162,816 B (3)
static_assert(kX6LdsBytes <= 160 * 1024, "LDS budget of one CU");
constexpr int kX6PiecesPerWave = kX6StageChunks / 4;                   // 12
__device__ __forceinline__ int x6_slot(int g) { return kX6Slots == 2 ? (g & 1) : g % 3; }

struct Ctx6 {
  const char* stream;   // weight stream in global memory
  char* lds;
  const char* buf;      // this lane's read base inside the current stage
  int g, G;             // next stage to begin, number of stages
  int wave_s, lane;
  // a workgroup that walks several point tiles (field_fwd_x6w.hip): the stream wraps -- the stage behind the last one
  // is stage 0 of the next tile -- and `par` is the number of stages of the tiles before, modulo the slot count
  int par = 0;
  bool wrap = false;
};
__device__ __forceinline__ int x6_slot(const Ctx6& c, int g) { return x6_slot(g + c.par); }

// each wave DMA-copies 12 of a stage's 48 chunks
__device__ __forceinline__ void x6_load_stage(const Ctx6& c, int g) {
  const char* src = c.stream + (long)g * kX6StageBytes + (c.wave_s * kX6PiecesPerWave) * 1024;
  char* dst = c.lds + x6_slot(c, g) * kX6StageBytes + (c.wave_s * kX6PiecesPerWave) * 1024;
#pragma unroll
  for (int q = 0; q < kX6PiecesPerWave; ++q) glds16_u(src + q * 1024, (unsigned)c.lane * 16u, dst + q * 1024);
}

// Barrier that retires stage g (its DMA has landed, every wave is done with stage g-1); the pieces of stage g+1 are
// issued by x6_load_piece() behind the MFMA groups of stage g.
__device__ __forceinline__ void x6_stage_begin(Ctx6& c) {
  // this wave's LDS-DMA pieces of the stage (inline asm: not in the compiler's count).  Three slots without the
  // read-ahead: all but the 12 youngest vector-memory operations -- the pieces of the stage after this one
#if DMN_X6_ABL != 1 && DMN_X6_ABL != 3       // timing-only ablations (results are wrong): 1 no wait for the stage's DMA,
  if (kX6Slots == 3 && !kX6CrossBarrier)   // 2 no DMA behind the first stages, 3 neither wait nor barrier
    asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
  else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
#if DMN_X6_ABL != 3
  __syncthreads();
#endif
  c.buf = c.lds + x6_slot(c, c.g) * kX6StageBytes + c.lane * 16;
  ++c.g;
}

// piece Q (0..11) of this wave's share of the stage kX6Slots - 1 ahead of the running one (the running stage is c.g - 1
// after x6_stage_begin); past the end of the stream the source clamps to the last stage (the copy lands in a slot nobody
// reads any more).  The two bases are per-stage values (one scalar expression each per stage, common to its 12 pieces);
// the piece's own offset is added inside the asm statement (glds16_at).
template <int Q, bool NT>
__device__ __forceinline__ void x6_load_piece(const Ctx6& c) {
#if DMN_X6_ABL == 2
  return;
#endif
  const int gt = c.g + kX6Slots - 2;
  const int g = gt < c.G ? gt : (c.wrap ? gt - c.G : c.G - 1);
  const char* src = c.stream + ((unsigned)g * (unsigned)kX6StageBytes + (unsigned)(c.wave_s * kX6PiecesPerWave * 1024));
  const unsigned dst = (unsigned)(unsigned long)(__attribute__((address_space(3))) char*)c.lds +
                       (unsigned)(x6_slot(c, gt) * kX6StageBytes + c.wave_s * kX6PiecesPerWave * 1024);
  glds16_at<Q * 1024, NT>(src, (unsigned)c.lane * 16u, dst);
}

// the same with the piece number as a (constant-folded) value
template <bool NT>
__device__ __forceinline__ void x6_piece(const Ctx6& c, int q) {
  switch (q) {
    case 0: x6_load_piece<0, NT>(c); break;
    case 1: x6_load_piece<1, NT>(c); break;
    case 2: x6_load_piece<2, NT>(c); break;
    case 3: x6_load_piece<3, NT>(c); break;
    case 4: x6_load_piece<4, NT>(c); break;
    case 5: x6_load_piece<5, NT>(c); break;
    case 6: x6_load_piece<6, NT>(c); break;
    case 7: x6_load_piece<7, NT>(c); break;
    case 8: x6_load_piece<8, NT>(c); break;
    case 9: x6_load_piece<9, NT>(c); break;
    case 10: x6_load_piece<10, NT>(c); break;
    default: x6_load_piece<11, NT>(c); break;
  }
}

// Exact three-way split of a value pair into elements j, j+1 of the hi / mid / lo fragments: one packed conversion per
// plane (v_cvt_pk_bf16_f32, round to nearest even), the rounded values back from the packed register by shift / mask,
// residuals by exact subtraction (x - bf16(x) has at most 16 significant bits, the second residual at most 8).
__device__ __forceinline__ void split3_pair(float x0, float x1, frag& fh, frag& fm, frag& fl, int j) {
  typedef P6::pair pair;
  typedef P6::float2v float2v;
  const float2v x = {x0, x1};
  const pair hp = __builtin_convertvector(x, pair);
#if DMN_X6_SCALAR_SUB   // four v_sub_f32 instead of two v_pk_add_f32 (a packed-f32 instruction beside MFMAs costs ~13 cycles more)
  const float2v wh = P6::widen(hp);
  float a0, a1, b0, b1;
  asm("v_sub_f32 %0, %1, %2" : "=v"(a0) : "v"(x0), "v"(wh[0]));
  asm("v_sub_f32 %0, %1, %2" : "=v"(a1) : "v"(x1), "v"(wh[1]));
  const float2v r1 = {a0, a1};
  const pair mp = __builtin_convertvector(r1, pair);
  const float2v wm = P6::widen(mp);
  asm("v_sub_f32 %0, %1, %2" : "=v"(b0) : "v"(a0), "v"(wm[0]));
  asm("v_sub_f32 %0, %1, %2" : "=v"(b1) : "v"(a1), "v"(wm[1]));
  const float2v r2 = {b0, b1};
  const pair lp = __builtin_convertvector(r2, pair);
#else
  const float2v r1 = x - P6::widen(hp);
  const pair mp = __builtin_convertvector(r1, pair);
  const float2v r2 = r1 - P6::widen(mp);
  const pair lp = __builtin_convertvector(r2, pair);
#endif
  fh[j] = hp[0]; fh[j + 1] = hp[1];
  fm[j] = mp[0]; fm[j + 1] = mp[1];
  fl[j] = lp[0]; fl[j + 1] = lp[1];
}

// ---- where a layer's input values come from: value of activation register i (feature feat_of(i, h)) ----------
// The accumulator tiles of the layer before (register i = tile i >> 4, element i & 15), read where the value is used by
// an explicit v_accvgpr_read (acc_read, field_x3.h): left to plain element reads the compiler copies the finished tiles
// out of their accumulator registers long before the k-step that consumes them and spills the copies.  Inline asm gets
// no MFMA -> VALU wait states from the compiler; none are needed: tile t of a layer's output is last written in that
// layer's last k-step, (NT - 1 - t) x 6 MFMAs before the layer ends, and first read by k-step 2t (- 1, one k-step
// ahead) of the next layer -- for t = NT - 1 that is at least 6 x NT x (2 NT - 3) MFMAs later, for t = 0 (read at the
// very start of the next layer) 6 (NT - 1) MFMAs = 576 cycles at NT = 4.  Code that reads a tile right behind its last
// MFMA (the colour head, the density-only exit) uses plain reads.
struct SrcTiles {
  const f32x16* a;
  __device__ __forceinline__ float operator()(int i) const { return acc_read(a[i >> 4], i & 15); }
};
struct SrcRegs {           // an encoding register file (pe / qe)
  const float* p;
  __device__ __forceinline__ float operator()(int i) const { return p[i]; }
};
// h8 = relu(mlps.7) for its second reader (ins_feature_linear): registers 0..63 still in their accumulator tiles,
// 64..127 activated in vector registers (see the kernel)
struct SrcH8 {
  const f32x16* a;
  const float* hi;
  __device__ __forceinline__ float operator()(int i) const {
    return i < 64 ? relu_bits(acc_read(a[(i & 63) >> 4], i & 15)) : hi[i < 64 ? 0 : i - 64];
  }
};

// ---- training: what a layer's input values leave behind in the save record as they pass by (field_layout.h "Training";
// the same record the exact-fp32 kernel writes, so the backward kernels do not care which forward ran).  Every
// activation is read out of its accumulator exactly once per consuming layer, by the operand split: that is where it is
// stored -- one row store per value, spread over the k-steps, and for post-ReLU layers one sign bit into the layer's
// mask words (bit b of word w <-> register 32 w + b).
struct NoSave {
  __device__ __forceinline__ void operator()(int, float, float) const {}
};
template <bool MASK>     // MASK: a post-ReLU layer (its sign bits are collected), else no activation behind the values
struct SaveRows {
  float* rows;           // record rows of this layer's values: register i -> rows + i * 64 (lane offset included)
  unsigned* msk;         // the kernel's four sign-bit words
  __device__ __forceinline__ void operator()(int i, float x0, float x1) const {
    record_store(rows + i * 64, x0);
    record_store(rows + (i + 1) * 64, x1);
    if (MASK) {          // post-ReLU values are +0 or positive: the sign bit is "bits != 0"
      const unsigned b0 = __float_as_uint(x0) != 0u ? 1u : 0u, b1 = __float_as_uint(x1) != 0u ? 1u : 0u;
      msk[i >> 5] |= (b0 << (i & 31)) | (b1 << ((i + 1) & 31));
    }
  }
};
template <int N>
__device__ __forceinline__ void x6_flush_mask(float* dst, unsigned* msk) {
#pragma unroll
  for (int w = 0; w < N; ++w) {
    record_store(dst + w * 64, __uint_as_float(msk[w]));
    msk[w] = 0u;
  }
}

// One layer, k-steps outer: out[t] (t < NT) = bias + sum over the NKS k-steps of `src` (ReLU applied first if RELU) and
// the NKX k-steps of `ext` (encodings, never activated) of the six-term product with the stream's [hi, mid, lo] chunks.
//   KPS    k-steps per 48-chunk stage (field_layout.h)
//   KEEP   the activated values of registers 64..127 stay behind in keep[i - 64] (they are in vector registers for the
//          split anyway; the caller reads them again later and gets their four accumulator tiles back)
//   LAST   the layer ends the weight stream: no LDS-DMA is issued during its last stage
//   sav    training: stores the (activated) values of `src` into the save record as they pass (NoSave: inference)
template <int LID, int NT, int NKS, int NKX, int KPS, bool RELU, bool KEEP, bool LAST, class Src, class Ext, class Sav = NoSave>
__device__ __forceinline__ void layer_x6(Ctx6& c, f32x16* out, const float* bias, const Src& src, const Ext& ext,
                                         float* keep = nullptr, const Sav& sav = Sav()) {
  constexpr int NK = NKS + NKX;
  constexpr bool NTP = ((DMN_X6_NTMASK >> LID) & 1) != 0;      // this layer's DMA pieces are non-temporal
  constexpr int GPS = KPS * NT;                 // (k-step, tile) groups per full stage
  static_assert(GPS * 3 <= kX6StageChunks, "a stage holds three chunks per group");
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int v = 0; v < 4; ++v) {              // bias = initial accumulator
      const f32x4 b = *reinterpret_cast<const f32x4*>(bias + t * 16 + v * 4);
      out[t][4 * v + 0] = b.x; out[t][4 * v + 1] = b.y; out[t][4 * v + 2] = b.z; out[t][4 * v + 3] = b.w;
    }
  // value pair p (0..3) of k-step ks -> elements 2p, 2p+1 of its operand fragments
  auto build_pair = [&](int ks, int p, frag& fh, frag& fm, frag& fl) {
    float x0, x1;
    if (ks < NKS) {
      x0 = src(8 * ks + 2 * p);
      x1 = src(8 * ks + 2 * p + 1);
      if (RELU) {
        x0 = relu_bits(x0);
        x1 = relu_bits(x1);
      }
      sav(8 * ks + 2 * p, x0, x1);
      if (KEEP && 8 * ks + 2 * p >= 64) {
        keep[8 * ks + 2 * p - 64] = x0;
        keep[8 * ks + 2 * p + 1 - 64] = x1;
      }
    } else {
      x0 = ext(8 * (ks - NKS) + 2 * p);
      x1 = ext(8 * (ks - NKS) + 2 * p + 1);
    }
    split3_pair(x0, x1, fh, fm, fl, 2 * p);
  };
  frag bh, bm, bl, nh, nm, nl;
#pragma unroll
  for (int p = 0; p < 4; ++p) build_pair(0, p, bh, bm, bl);
  frag ah[2], am[2], al[2];                     // A fragments of the running group and of the next one
#pragma unroll
  for (int ks = 0; ks < NK; ++ks) {
    constexpr int kLastStage = (NK - 1) / KPS;
    const int stage = ks / KPS;
    const int groups = (stage == kLastStage ? NK - stage * KPS : KPS) * NT;     // groups of this stage (last: maybe fewer)
    // with three slots every stage issues its 12 pieces (clamped past the end): the counted wait relies on it
    const bool dma = kX6Slots == 3 || !(LAST && stage == kLastStage);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const int gi = (ks % KPS) * NT + t;       // group inside its stage
      const int ar = (ks * NT + t) & 1;
      if (gi == 0) {
        x6_stage_begin(c);
        if (!(kX6CrossBarrier && stage > 0)) {  // (read ahead of the barrier by the previous stage's last group)
          ah[ar] = *reinterpret_cast<const frag*>(c.buf);
          am[ar] = *reinterpret_cast<const frag*>(c.buf + 1024);
          al[ar] = *reinterpret_cast<const frag*>(c.buf + 2048);
        }
      }
      if (gi + 1 < groups) {                    // next group's fragments fly under this group's MFMAs
        ah[ar ^ 1] = *reinterpret_cast<const frag*>(c.buf + (3 * (gi + 1)) * 1024);
        am[ar ^ 1] = *reinterpret_cast<const frag*>(c.buf + (3 * (gi + 1) + 1) * 1024);
        al[ar ^ 1] = *reinterpret_cast<const frag*>(c.buf + (3 * (gi + 1) + 2) * 1024);
      } else if (kX6CrossBarrier && stage < kLastStage) {      // the next stage landed one barrier ago: read ahead
        const char* nb = c.lds + x6_slot(c.g) * kX6StageBytes + c.lane * 16;
        ah[ar ^ 1] = *reinterpret_cast<const frag*>(nb);
        am[ar ^ 1] = *reinterpret_cast<const frag*>(nb + 1024);
        al[ar ^ 1] = *reinterpret_cast<const frag*>(nb + 2048);
      }
      __builtin_amdgcn_sched_barrier(0);        // keep the prefetch above this group's MFMAs
#if DMN_X6_NINE                                 // experiment: all nine products (the three dropped ones first)
      out[t] = P6::mfma(al[ar], bl, out[t]);
      out[t] = P6::mfma(al[ar], bm, out[t]);
      out[t] = P6::mfma(am[ar], bl, out[t]);
#endif
#if DMN_X6_ORDER == 0
      out[t] = P6::mfma(al[ar], bh, out[t]);    // small terms first
      out[t] = P6::mfma(ah[ar], bl, out[t]);
      out[t] = P6::mfma(am[ar], bm, out[t]);
      out[t] = P6::mfma(am[ar], bh, out[t]);
      out[t] = P6::mfma(ah[ar], bm, out[t]);
      out[t] = P6::mfma(ah[ar], bh, out[t]);
#else                                           // consecutive products share an operand (experiment)
      out[t] = P6::mfma(ah[ar], bl, out[t]);
      out[t] = P6::mfma(ah[ar], bm, out[t]);
      out[t] = P6::mfma(am[ar], bm, out[t]);
      out[t] = P6::mfma(am[ar], bh, out[t]);
      out[t] = P6::mfma(al[ar], bh, out[t]);
      out[t] = P6::mfma(ah[ar], bh, out[t]);
#endif
      if (dma) {
        if (gi < kX6PiecesPerWave) x6_piece<NTP>(c, gi);
        if (gi + 1 == groups) {                 // a short stage still owes the rest of its pieces
#pragma unroll
          for (int q = gi + 1; q < kX6PiecesPerWave; ++q) x6_piece<NTP>(c, q);
        }
      }
      if (ks + 1 < NK) {                        // the next k-step's operand fragments, a pair (or more) per tile
#pragma unroll
        for (int p = 0; p < 4; ++p)
          if ((p * NT) / 4 == t) build_pair(ks + 1, p, nh, nm, nl);
      }
    }
    bh = nh;
    bm = nm;
    bl = nl;
  }
}

// The whole network on this wave's 32 points (one per lane pair; lane-half h = lane >> 5), networks/dm_nerf.py:80-106.
//   c      weight stream context (stage 0 -- and 1 with three slots -- already requested by the caller)
//   cst    constant section in LDS;  pe / qe: the encoding registers of point_encodings()
//   rec    SAVE (training): this wave's save record, lane offset included (field_layout.h "Training")
// Outputs (valid in both lane halves): sigma, rgb[3], and insv[i] = instance logit of channel feat_of(i, 0) + 4 h
// (i < 16 T).  With sigma_only the network stops behind its density head (the colour and instance branches are 29 % of
// it) and returns true: sigma is valid, rgb / insv are not written.
template <int T, bool SAVE>
__device__ __forceinline__ bool x6_network(Ctx6& c, const float* cst, const float* pe, const float* qe, int h,
                                           bool sigma_only, float* rec, float& sigma, float (&rgb)[3],
                                           float (&insv)[16 * T]) {
  unsigned msk[4] = {0u, 0u, 0u, 0u};
  f32x16 accA[8], accB[8];
  sigma = 0.f;
  const float* const dw = cst + kConstDensW + h * 128;
  const SrcRegs s_pe{pe}, s_qe{qe};
  const SrcTiles s_a{accA}, s_b{accB};
  constexpr int K8 = x6_ksteps_per_stage(8), K4 = x6_ksteps_per_stage(4);

  const_section_ready();      // mlps.0 reads its bias before its first stage barrier
  // ---- mlps.0: position encoding (4 k-steps) -> A
  layer_x6<0, 8, 4, 0, K8, false, false, false>(c, accA, cst + kConstBias8 + h * 128, s_pe, s_pe);
  // ---- mlps.1 .. mlps.4 in pairs A -> B -> A (rolled: a layer is ~2500 instructions); training: h_l = relu(mlps.(l-1))
  //      goes into the record while mlps.l consumes it
#pragma unroll 1
  for (int l = 1; l < 5; l += 2) {
    if constexpr (SAVE) {
      layer_x6<1, 8, 16, 0, K8, true, false, false>(c, accB, cst + kConstBias8 + l * 256 + h * 128, s_a, s_pe, nullptr,
                                                 SaveRows<true>{rec + save_h(l) * 64, msk});
      x6_flush_mask<4>(rec + save_mask(l) * 64, msk);
      layer_x6<1, 8, 16, 0, K8, true, false, false>(c, accA, cst + kConstBias8 + (l + 1) * 256 + h * 128, s_b, s_pe, nullptr,
                                                 SaveRows<true>{rec + save_h(l + 1) * 64, msk});
      x6_flush_mask<4>(rec + save_mask(l + 1) * 64, msk);
    } else {
      layer_x6<1, 8, 16, 0, K8, true, false, false>(c, accB, cst + kConstBias8 + l * 256 + h * 128, s_a, s_pe);
      layer_x6<1, 8, 16, 0, K8, true, false, false>(c, accA, cst + kConstBias8 + (l + 1) * 256 + h * 128, s_b, s_pe);
    }
  }
  // ---- mlps.5 on cat([h, input_pts]) (skip connection, dm_nerf.py:86-87), mlps.6, mlps.7: A -> B -> A -> B
  // (c.g = ...: the stage number as a constant again behind the rolled loop -- the running counter is the same number,
  // but as a loop-carried value it makes every later LDS-DMA address a scalar expression of it, which the compiler
  // precomputes behind the loop, 324 of them, and parks in vector-register lanes: 580 v_writelane / v_readlane per pass)
  c.g = kX6StageL5;
  if constexpr (SAVE) {
    layer_x6<2, 8, 16, 4, K8, true, false, false>(c, accB, cst + kConstBias8 + 5 * 256 + h * 128, s_a, s_pe, nullptr,
                                               SaveRows<true>{rec + save_h(5) * 64, msk});
    x6_flush_mask<4>(rec + save_mask(5) * 64, msk);
    c.g = kX6StageL6;
    layer_x6<3, 8, 16, 0, K8, true, false, false>(c, accA, cst + kConstBias8 + 6 * 256 + h * 128, s_b, s_pe, nullptr,
                                               SaveRows<true>{rec + save_h(6) * 64, msk});
    x6_flush_mask<4>(rec + save_mask(6) * 64, msk);
    c.g = kX6StageL6 + 8;
    layer_x6<4, 8, 16, 0, K8, true, false, false>(c, accB, cst + kConstBias8 + 7 * 256 + h * 128, s_a, s_pe, nullptr,
                                               SaveRows<true>{rec + save_h(7) * 64, msk});
    x6_flush_mask<4>(rec + save_mask(7) * 64, msk);
  } else {
    layer_x6<2, 8, 16, 4, K8, true, false, false>(c, accB, cst + kConstBias8 + 5 * 256 + h * 128, s_a, s_pe);
    c.g = kX6StageL6;
    layer_x6<3, 8, 16, 0, K8, true, false, false>(c, accA, cst + kConstBias8 + 6 * 256 + h * 128, s_b, s_pe);
    c.g = kX6StageL6 + 8;
    layer_x6<4, 8, 16, 0, K8, true, false, false>(c, accB, cst + kConstBias8 + 7 * 256 + h * 128, s_a, s_pe);
  }

  // ---- density_linear on h8 = relu(B) (dm_nerf.py:101): 256 -> 1 on the VALU.  (Taken while h8 passes through the
  //      next layer's operand split instead, its 128 weight reads are hoisted together and cost 100 registers there.)
#pragma unroll
  for (int i = 0; i < 128; ++i) sigma = fmaf(relu_bits(accB[i >> 4][i & 15]), dw[i], sigma);
  sigma += __shfl_xor(sigma, 32);
  sigma += cst[kConstDensB];
  // density is all the caller wants (the coarse pass of an image loop feeds the resampler): out, with rgb / insv
  // untouched.  A run-time branch that LEAVES (the stand-alone kernel stores its row and ends there): as a compile-time
  // variant, or with the two paths joining again behind the network, the compiler treats the 6,600 MFMAs as one
  // region, stretches live ranges across it and spills 100+ vector registers.
  if (sigma_only) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the next stage's LDS-DMA must land before the LDS is let go of
    return true;
  }
  rgb[0] = 0.f; rgb[1] = 0.f; rgb[2] = 0.f;

  // ---- rgb_feature_linear on h8 (no activation behind it, dm_nerf.py:89) -> A.  The upper half of h8 (tiles 4..7 of
  //      B) stays behind in vector registers (the position encoding's are free by now): its four tiles take the output
  //      of the 4-tile layers, so 2 x 128 accumulator registers stay the maximum; ins_feature_linear reads h8 from
  //      both places
  float h8[64];
  f32x16 acc2[4];
  const float* const rw = cst + kConstRgbW + h * 64;
  const SrcH8 s_h8{accB, h8};
  const SrcTiles s_2{acc2};
  f32x16 acc3[T];
  c.g = kX6StageRgbF;
  if constexpr (SAVE) {
    layer_x6<5, 8, 16, 0, K8, true, true, false>(c, accA, cst + kConstBias8 + 8 * 256 + h * 128, s_b, s_pe, h8,
                                              SaveRows<true>{rec + save_h(8) * 64, msk});
    x6_flush_mask<4>(rec + save_mask(8) * 64, msk);
    // ---- rgb_feature_linears.0 on cat([feature, dirs]) -> 128, ReLU (dm_nerf.py:90-93); rgb_linear on the VALU (:102)
    c.g = kX6StageRgbH;
    layer_x6<6, 4, 16, 2, K4, false, false, false>(c, acc2, cst + kConstBias4 + h * 64, s_a, s_qe, nullptr,
                                                SaveRows<false>{rec + kSaveFeat * 64, msk});
  } else {
    layer_x6<5, 8, 16, 0, K8, true, true, false>(c, accA, cst + kConstBias8 + 8 * 256 + h * 128, s_b, s_pe, h8);
    c.g = kX6StageRgbH;
    layer_x6<6, 4, 16, 2, K4, false, false, false>(c, acc2, cst + kConstBias4 + h * 64, s_a, s_qe);
  }
#pragma unroll
  for (int i0 = 0; i0 < 64; i0 += 8) {         // eight values at a time: h8 and the view encoding stay in registers beside it
#pragma unroll
    for (int i = i0; i < i0 + 8; i += 2) {
      const float x0 = relu_bits(acc2[i >> 4][i & 15]), x1 = relu_bits(acc2[(i + 1) >> 4][(i + 1) & 15]);
      if (SAVE) SaveRows<true>{rec + kSaveR1 * 64, msk}(i, x0, x1);
#pragma unroll
      for (int k = 0; k < 3; ++k) rgb[k] = fmaf(x1, rw[k * 128 + i + 1], fmaf(x0, rw[k * 128 + i], rgb[k]));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
  if (SAVE) x6_flush_mask<2>(rec + kSaveMaskR1 * 64, msk);
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    rgb[k] += __shfl_xor(rgb[k], 32);
    rgb[k] += cst[kConstRgbB + k];
  }
  // ---- ins_feature_linear on the same trunk output (h.detach(), dm_nerf.py:95-96), no activation -> A;
  //      ins_feature_linears.0 -> 128, ReLU (:97-99); ins_linear -> ins_num + 1 (:103)
  c.g = kX6StageInsF;
  layer_x6<7, 8, 16, 0, K8, false, false, false>(c, accA, cst + kConstBias8 + 9 * 256 + h * 128, s_h8, s_pe);
  c.g = kX6StageInsH;
  if constexpr (SAVE) {
    layer_x6<8, 4, 16, 0, K4, false, false, false>(c, acc2, cst + kConstBias4 + 128 + h * 64, s_a, s_pe, nullptr,
                                                SaveRows<false>{rec + kSaveInsF * 64, msk});
    c.g = kX6StageInsO;
    layer_x6<9, T, 8, 0, x6_inso_ksteps_per_stage(T), true, false, true>(c, acc3, cst + kConstBiasO + h * 64, s_2, s_pe, nullptr,
                                                                     SaveRows<true>{rec + kSaveI1 * 64, msk});
    x6_flush_mask<2>(rec + kSaveMaskI1 * 64, msk);
  } else {
    layer_x6<8, 4, 16, 0, K4, false, false, false>(c, acc2, cst + kConstBias4 + 128 + h * 64, s_a, s_pe);
    c.g = kX6StageInsO;
    layer_x6<9, T, 8, 0, x6_inso_ksteps_per_stage(T), true, false, true>(c, acc3, cst + kConstBiasO + h * 64, s_2, s_pe);
  }

#pragma unroll
  for (int i = 0; i < 16 * T; ++i) insv[i] = acc3[i >> 4][i & 15];
  return false;
}

}  // namespace dmn
