// Launch parameters shared by the flash-attention kernels (attention_fwd.hip, attention_w64.hip).
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
  int q_prescaled;
  int Bkv;    // key / value batches: query batch b reads key batch b % Bkv
  int map;    // block -> (head, query block) order: 0 = heads fastest, 1 = XCD-contiguous, query blocks fastest
};

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

// attention_w64.hip: 4 waves x 64 query rows, one wave per SIMD, LDS-DMA staged K / V^T
int launch_w64(const AttnParams& p, int hd, int hints, hipStream_t st);
// attention_asm72.hip: the same structure for head_dim 72 with a hand-scheduled (generated) main loop
bool asm72_supported(const AttnParams& p, int hd);
int launch_asm72(const AttnParams& p, int nu, int var, hipStream_t st);
// attention_asm128.hip: head_dim 128, 4 waves x 64 rows, generated main loop
int launch_asm128(const AttnParams& p, int var, hipStream_t st);

}  // namespace osk_attn
