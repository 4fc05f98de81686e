// Arguments of the matrix-core attention kernels (attention_mfma.hip: exact fp32 MFMA; attention_planes.hip: fp32-grade on
// the bf16 matrix cores) and the entry points attention.hip dispatches to.
#pragma once
#include "msn_common.h"

namespace msn {

struct MAttn {
    const float* q; const float* k; const float* v; const float* o; const float* dout;
    float* out; float* dq; float* dk; float* dv;
    const uint8_t* mask;      // [B][Tk] or null
    float* lse;               // [B][H][Tq][2] = (row max, log sum)
    float* delta;             // [B][H][Tq]
    int64_t ldq, ldk, ldv, ldo, ldd, lddq, lddk, lddv;
    int64_t q_bs, k_bs, v_bs, o_bs, d_bs, dq_bs, dk_bs, dv_bs;
    int B, H, Tq, Tk, hd;
    float scale;
    int tail;                 // set by mattn_forward / mattn_backward: ragged last token on the vector ALU
    // forward only (msn_attention_fwd_planes): the output ALSO as a plane matrix of (B Tq) rows x (H hd) columns -- the operand
    // of the output projection -- written by the kernel that computes it instead of a split pass; null: off
    unsigned char* oplanes;
    int o_np, o_cb;           // planes (2 | 3), column blocks of the plane matrix
};

// shared_q: also accept q_bs == 0 (one query shared by the batch: every kernel reads Q at q + b * q_bs and writes dq per sample)
bool mattn_applicable(const MAttn& a, bool shared_q = false);
int mattn_forward(const MAttn& a, hipStream_t st);
int mattn_backward(const MAttn& a, hipStream_t st);

// attention_planes.hip: heads up to 16 wide, any sequence length
bool pattn_applicable(const MAttn& a);
int pattn_forward(const MAttn& a, hipStream_t st);
int pattn_backward(const MAttn& a, hipStream_t st);
bool pattn_backward_preferred(const MAttn& a);   // heads narrower than 16 whose BACKWARD is faster on the planes (forward: vector ALU)

// (sample, head, row block) of workgroup blockIdx.x; with B % 8 == 0 the H * NB workgroups of a sample are consecutive
// workgroups of ONE XCD (ids go to the XCDs round-robin) and share the 128-byte lines of its q|k|v rows in that L2
__device__ __forceinline__ void locate_block(const MAttn& p, int NB, int& b, int& hh, int& blk) {
    const unsigned per = (unsigned)(p.H * NB);
    unsigned id = blockIdx.x;
    if ((p.B & 7) == 0) {
        const unsigned xcd = id & 7, j = id >> 3;
        id = ((j / per) * 8 + xcd) * per + j % per;
    }
    blk = (int)(id % NB);
    const unsigned bh = id / NB;
    b = (int)(bh / p.H), hh = (int)(bh % p.H);
}

}  // namespace msn
