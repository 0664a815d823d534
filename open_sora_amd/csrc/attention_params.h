// Launch parameters shared by the flash-attention kernels (attention_fwd.hip, attention_asm{72,128}{,p8}.hip).
#pragma once
#include "osk_common.h"

namespace osk_attn {

struct AttnParams {
  const unsigned short* q;
  int64_t qbs, qrs;
  const unsigned short* k;
  int64_t kss, kbs, krs;
  const unsigned short* vt;
  int64_t vtss;
  unsigned short* out;
  int64_t obs, ors;
  float* lse;
  int B, H, Lq, n_seg, seg_len, seg_lp, tps;
  float sc;   // softmax scale * log2(e); 1.0 when q_prescaled
  // caller-supplied bound on the scores as the kernel sees them (|q . k| * sc <= bound, log2 units), rounded UP to a bf16 value;
  // 0 = unknown.  With a bound <= OSK_ATTN_MAX_BOUND the hand-scheduled kernels run their FAST body: reference max = bound (a
  // constant in Q's padding dim), no max tracking, branch-free loader advance; ragged segment-last tiles and segment jumps are
  // out-of-line events (tools/gen_attn_asm.py::fast_events).  Several segments of fewer than 3 tiles: the general body.
  float bound = 0.f;
  int q_prescaled;
  int rows = 256;   // query rows per work unit (workgroup): 256, or 512 for the wide head_dim-72 layout (attention_asm72w.hip)
  int Bkv;    // key / value batches: query batch b reads key batch b % Bkv
  int map;    // block -> (head, query block) order: 0 = heads fastest, 1 = XCD-contiguous, query blocks fastest
  // tail split (hand-scheduled kernels only; see split_tail()): the workgroups of the last, partial round of the grid
  // -- work units >= tail_first in launch order -- are cut into tail_split parts along the key axis; a part writes its
  // normalised partial O (f32) and its log2-domain LSE into the workspace and attn_merge_kernel combines them
  int tail_first = 0x7fffffff, tail_split = 1;
  float* ws_o = nullptr;
  float* ws_lse = nullptr;
  // fp8 P.V variant (attention_asm*p8.hip): V^T as e4m3 [Bkv, H, RP, seg_lp] per segment (vtss in BYTES) from
  // osk_v_transpose_fp8, one f32 scale per (key batch, head)
  const unsigned char* vt8 = nullptr;
  const float* v_scale = nullptr;
  // device-derived score bound (round 6, osk_attention_fwd_auto_bf16): squared row-norm maxima of the q and k the kernel receives,
  // per (batch, head) -- qn2 [B, H], kn2 [Bkv, H], from osk_rownorm2_max_bf16.  With both set, every workgroup derives ITS bound
  // sqrt(qn2 kn2) (Cauchy-Schwarz on the actual operands) and the launch is one of a PAIR over the same grid: the FAST kernel's
  // workgroups run where that bound is <= OSK_ATTN_MAX_BOUND and exit at once elsewhere, the general kernel's the other way round
  // (attn_auto_bound below) -- no host round trip, no promise from the caller, hipGraph-capturable.
  const float* qn2 = nullptr;
  const float* kn2 = nullptr;
};

#define OSK_ATTN_MAX_BOUND 56.0f   // P = exp2(s - bound) >= 2^-112 for every admissible score: no underflow to zero

static inline bool attn_fast_path(const AttnParams& p) {   // host side
  return p.bound > 0.f && p.bound <= OSK_ATTN_MAX_BOUND && (p.n_seg == 1 || p.tps >= 3);
}

// Which of the two kernels of an auto-dispatched pair works on (batch b, head h), and with which bound?  Host-decided launches
// (qn2 == nullptr): every workgroup runs, bound = the caller's.  Returns false when THIS kernel (fast_kernel: its FAST instantiation)
// must leave the unit to its twin.  Wave-uniform (scalar loads).
OSK_DEV bool attn_auto_bound(const AttnParams& p, int b, int h, bool fast_kernel, float& bound) {
  bound = p.bound;
  if (!p.qn2) return true;
  // the norms are those of the bf16 values the MFMA multiplies: |q . k| <= |q| |k| exactly; 1.004 covers the f32 rounding of the
  // products and the square root; then UP to the next bf16 value (the kernels keep the bound in a bf16 field of Q's padding dim)
  const float v = __builtin_sqrtf(p.qn2[b * p.H + h] * p.kn2[(b % p.Bkv) * p.H + h]) * 1.004f;
  bound = __uint_as_float((__float_as_uint(v) + 0xFFFFu) & 0xFFFF0000u);
  const bool fast_ok = bound <= OSK_ATTN_MAX_BOUND;      // (NaN / inf norms: false -> the general body)
  return fast_ok == fast_kernel;
}

// (batch*head, query block) of a workgroup.  map 1 hands every XCD (block b runs on XCD b % 8) a contiguous
// range of the (head-major) work list, so the workgroups resident on one XCD walk the SAME head's K / V^T
// stream together and share it through that XCD's private L2.
OSK_DEV void block_to_work(const AttnParams& p, int nqb, int& bh, int& qb) {
  const int nbh = p.B * p.H;
  if (p.map == 1) {
    const int w = xcd_remap(blockIdx.x, nqb * nbh);
    bh = w / nqb;
    qb = w - bh * nqb;
  } else {
    bh = blockIdx.x % nbh;
    qb = blockIdx.x / nbh;
  }
}

// work unit index (launch order) -> (batch*head, query block)
OSK_DEV void unit_to_work(const AttnParams& p, int nqb, int unit, int& bh, int& qb) {
  const int nbh = p.B * p.H;
  if (p.map == 1) {
    const int w = xcd_remap(unit, nqb * nbh);
    bh = w / nqb;
    qb = w - bh * nqb;
  } else {
    bh = unit % nbh;
    qb = unit / nbh;
  }
}

// block -> work unit (+ key part of a split tail unit).  Returns true for a part of a split unit; tail_unit = its index
// among the tail units (workspace slot).
OSK_DEV bool block_to_work_split(const AttnParams& p, int nqb, int& bh, int& qb, int& part, int& tail_unit) {
  int unit = blockIdx.x;
  part = 0;
  tail_unit = 0;
  const bool tail = p.tail_split > 1 && unit >= p.tail_first;
  if (tail) {
    const int j = unit - p.tail_first;
    tail_unit = j / p.tail_split;
    part = j - tail_unit * p.tail_split;
    unit = p.tail_first + tail_unit;
  }
  unit_to_work(p, nqb, unit, bh, qb);
  return tail;
}

// key range of a part: element offsets of its first tile into K and V^T, its tile counts, whether it ends in the ragged tile
struct KeyPart {
  int64_t k_off, v_off;
  int tps, nt;
  bool ragged;
};
OSK_DEV KeyPart key_part(const AttnParams& p, bool tail, int part, bool ragged) {
  KeyPart r{0, 0, p.tps, p.n_seg * p.tps, ragged};
  if (tail) {
    if (p.n_seg > 1) {   // whole key segments per part (tail_split divides n_seg)
      const int per = p.n_seg / p.tail_split;
      r.k_off = (int64_t)part * per * p.kss;
      r.v_off = (int64_t)part * per * p.vtss;
      r.nt = per * p.tps;
    } else {             // a run of tiles of the single segment; only the last part ends in the (possibly ragged) last tile
      const int t0 = part * p.tps / p.tail_split, t1 = (part + 1) * p.tps / p.tail_split;
      r.k_off = (int64_t)t0 * 64 * p.krs;
      r.v_off = (int64_t)t0 * 64;
      r.tps = r.nt = t1 - t0;
      r.ragged = ragged && part == p.tail_split - 1;
    }
  }
  return r;
}

// host: decide the tail split of a launch of `units` work units given the workspace (attention_fwd.hip)
void split_tail(AttnParams& p, int units, int hd, void* workspace, int64_t workspace_bytes);
int launch_merge(const AttnParams& p, int hd, hipStream_t st);

// attention_asm72.hip: head_dim 72, 4 waves x 64 query rows, hand-scheduled (generated) main loop
int launch_asm72(const AttnParams& p, hipStream_t st);
// attention_asm72w.hip: head_dim 72, the bounded body in the wide layout (4 waves x 128 rows, one 32-key half per loop body); p.rows == 512
int launch_asm72w(const AttnParams& p, hipStream_t st);
// host: does a bounded head_dim-72 call take the wide layout?  (enough query rows for 512-row work units to fill the chip evenly)
// the same two kernels on tensors with 64-wide heads (template parameter HD = 64)
int launch_asm64(const AttnParams& p, hipStream_t st);
int launch_asm64w(const AttnParams& p, hipStream_t st);
// Rounds of the chip a launch of `units` equal work units takes, in units of one work unit's time: whole rounds plus the last,
// partial one -- which costs a full round without a workspace and min over the admissible key splits s of ceil(R s / CUs) / s with
// one (split_tail below makes the same choice).
static inline double attn_launch_rounds(const AttnParams& p, int units, int cus, bool has_ws) {
  const int R = units % cus;
  double tail = R ? 1.0 : 0.0;
  if (R && has_ws)
    for (int s = 2; s <= 8; ++s) {
      if (p.n_seg > 1 ? (p.n_seg % s != 0) : (s > p.tps)) continue;
      const double c = (double)((R * s + cus - 1) / cus) / s;
      if (c < tail) tail = c;
    }
  return (double)(units / cus) + tail;
}
// 512-row (wide) or 256-row work units for a bounded head_dim 72 / 64 call?  By estimated time, as tile_choice() does for the
// GEMMs (round 4 looked at Lq >= 1024 only: with few batch x head pairs -- B = 1, or a head-parallel sequence-parallel rank with
// H / P heads -- halving the unit count can leave most of the chip idle): a wide unit does the work of two narrow ones
// in 1.85 x the time (-7.4 % per launch measured where both fill the chip, profiles/r04e_attn_wide_step_ab.jsonl).
static inline bool attn_wide_path(const AttnParams& p, int hd, int cus, bool has_ws) {
#ifdef OSK_ATTN_NO_WIDE   // (A/B builds of tools/make_attn_nowide_lib.sh: always the 256-row layout)
  (void)p; (void)hd; (void)cus; (void)has_ws;
  return false;
#else
  if (!((hd == 72 || hd == 64) && attn_fast_path(p) && p.Lq >= 1024)) return false;
  const int bh = p.B * p.H;
  const double narrow = attn_launch_rounds(p, ((p.Lq + 255) / 256) * bh, cus, has_ws);
  const double wide = 1.852 * attn_launch_rounds(p, ((p.Lq + 511) / 512) * bh, cus, has_ws);
  return wide <= narrow;
#endif
}
// attention_asm128.hip: head_dim 128, the same layout and generator
int launch_asm128(const AttnParams& p, hipStream_t st);
// attention_asm128p8.hip / attention_asm72p8.hip: the same with the P.V product on the fp8 MFMA
int launch_asm128p8(const AttnParams& p, hipStream_t st);
int launch_asm72p8(const AttnParams& p, hipStream_t st);

}  // namespace osk_attn
