// Shared pieces of the bf16 MFMA GEMM kernels: argument block, epilogue selector and the fused epilogue.
#pragma once
#include "common.hpp"

enum { EPI_BF16 = 0, EPI_GELU = 1, EPI_DGELU = 2, EPI_F32 = 3 };
#define EPI_QKV_API 4   // C-ABI only: EPI_BF16 whose first N/3 output columns (the q part of a qkv projection) are multiplied by alpha

struct GemmArgs {
  const bf16_t* A;
  const bf16_t* B;
  void* C;
  const float* bias;      // [N] fp32, nullable
  const bf16_t* res;      // [M,N] bf16 residual, nullable (EPI_BF16)
  const bf16_t* aux_in;   // [M,N] bf16 saved GELU derivative gelu'(u) (EPI_DGELU): what EPI_GELU wrote to aux_out
  bf16_t* aux_out;        // [M,N] bf16 gelu'(u) of the bf16-rounded pre-activation u = acc + bias, nullable (EPI_GELU)
  int64_t M, N, K, lda, ldb, ldc, ldr, ldaux;
  float alpha, beta;
  int tiles_m, tiles_n;
  int splitk;          // > 1: K is cut into `splitk` slices, raw fp32 partials go to ws[slice][M][N] (EPI_F32 only)
  int ktiles_per;      // K-tiles per slice
  float* ws;
  const bf16_t* zero_row;   // 128 bf16 zeros: source of token rows beyond T in the TN weight-gradient kernel (gemm8_tn.hip)
  int dbg;             // diagnostics (env VJ_GEMM_DBG, tools/gemm_ksweep.py): bit0 = drop the epilogue, bit1 = direct (unstaged) stores
  float qscale;        // EPI_BF16 without residual: != 0 -> columns n < qcols are multiplied by qscale before the bf16 rounding (the q part
  int64_t qcols;       //   of a qkv projection carries the soft-max scale scale*log2(e): ONE rounding of c*q, attention.hip); qcols % 4 == 0
  float* colpart;      // EPI_DGELU on the persistent kernel, nullable: fp32 column-sum partials of the OUTPUT, [2 * tiles_m][N]
                       // (row slot = 2 * row tile + wave row): the bias gradient of the Linear whose dY this GEMM produces
  int raster;          // persistent kernel: tile order (tile_of_raster; 0 = the default 8-row groups)
  // LayerNorm folded into this GEMM (round 5; bf16 / GELU epilogues without residual or saved derivative): A holds the RAW rows x
  // (not LayerNorm(x)), B = bf16(W * diag(gamma)), bias = b + W beta, lnf_c[n] = sum_k B[n,k] and lnf_rs[m] = {rstd_m, -mean_m * rstd_m}:
  //   out[m,n] = rstd_m * (acc[m,n] - mean_m * c[n]) + bias[n]  =  LayerNorm(x)[m,:] . W[n,:] + b[n]      (vj_gemm_bf16_nt_lnfold)
  const float* lnf_rs;   // [M][2] fp32, nullable (null: plain epilogue)
  const float* lnf_c;    // [N] fp32
  int half_tiles = 0;    // persistent kernel: N % 256 == 128 and the shifted last column tile computes its own 128 columns only (gemm8p.hip)
  int epi_pre = 0;       // persistent kernel: form of the epilogue (option gemm_epi_pre: 4 = pipelined passes, 0 = straight passes; PRE below)
};


// Fused epilogue.  acc[i][j] comes from MFMA 16x16x32 issued with swapped operands (D = Bfrag x Afrag): lane
// (g = lane>>4, r = lane&15) owns row m = .. + i*16 + r and the 4 consecutive columns n = .. + j*16 + 4g + {0..3}.
//
// The body is deliberately STRAIGHT-LINE code: on gfx9 stores count in vmcnt together with loads, and the compiler's
// wait-count insertion falls back to `s_waitcnt vmcnt(0)` -- i.e. "drain every store issued so far" -- at each use of a
// loaded value once control flow separates the load from the use.  So there are no `continue`s and no per-block
// branches here: edge rows / columns are handled by clamping load addresses and predicating the stores only, the
// nullable operands are resolved once by the caller-side variant switch (HAS_OPT), the bias is complete before the
// first row (one explicit wait), and the row operands (residual / saved pre-activation) are fetched one row-block ahead
// so that a row's stores stay in flight while the next row is computed.
template <int EPI, int FM, int FN, bool HAS_OPT, bool EDGE, bool BETA = false, bool QS = false, bool LP = false, bool LNF = false>
__device__ __forceinline__ void gemm_epilogue_impl(const GemmArgs& p, f32x4_t (&acc)[FM][FN], int64_t m_base,
                                                   int64_t n_base, int frow, int fg, int slice) {
  static_assert(!QS || (EPI == EPI_BF16 && !HAS_OPT), "column scale: the qkv projection (bf16 output, no residual)");
  static_assert(!LNF || ((EPI == EPI_BF16 || EPI == EPI_GELU) && !HAS_OPT), "LayerNorm fold: bf16 / GELU epilogue, no residual, no saved derivative");
  const int64_t ncol0 = n_base + fg * 4;   // this lane's first column; tile j adds j*16
  f32x2_t qs2[QS ? FN : 1];
  if constexpr (QS) {
#pragma unroll
    for (int j = 0; j < FN; j++) {
      const float sj = (ncol0 + j * 16 < p.qcols) ? p.qscale : 1.0f;
      qs2[j] = (f32x2_t){sj, sj};
    }
  }
  bool cok[FN];
  int64_t ncl[FN];                         // column for loads, clamped into the matrix (N % 4 == 0, N >= 4)
#pragma unroll
  for (int j = 0; j < FN; j++) {
    const int64_t n = ncol0 + j * 16;
    cok[j] = EDGE ? n < p.N : true;
    ncl[j] = cok[j] ? n : 0;
  }
  float4 bias4[FN];
#pragma unroll
  for (int j = 0; j < FN; j++) bias4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
  if constexpr (EPI != EPI_F32) {
    if (p.bias) {
#pragma unroll
      for (int j = 0; j < FN; j++) bias4[j] = *(const float4*)(p.bias + ncl[j]);
    }
  }
  float4 lc4[LNF ? FN : 1];
  if constexpr (LNF) {
#pragma unroll
    for (int j = 0; j < FN; j++) lc4[j] = *(const float4*)(p.lnf_c + ncl[j]);
  }
  __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0), visible to the compiler: nothing older than the epilogue is pending

  if constexpr (EPI == EPI_F32) {
    // HAS_OPT = split-K: raw partials into ws[slice]; alpha / beta are applied by the slice reduction
#pragma unroll
    for (int i = 0; i < FM; i++) {
      const int64_t m = m_base + i * 16 + frow;
      const bool mok = EDGE ? m < p.M : true;
      const int64_t mc = mok ? m : p.M - 1;
      float* c32 = (float*)p.C + mc * p.ldc;
      float* wsp = p.ws + ((int64_t)slice * p.M + mc) * p.N;
#pragma unroll
      for (int j = 0; j < FN; j++) {
        float4 o = make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        if constexpr (HAS_OPT) {
          if (mok && cok[j]) *(float4*)(wsp + ncl[j]) = o;
        } else {
          o.x *= p.alpha;
          o.y *= p.alpha;
          o.z *= p.alpha;
          o.w *= p.alpha;
          if constexpr (BETA) {   // accumulate into C (not used on the training path: every wgrad overwrites)
            const float4 c0 = *(const float4*)(c32 + ncl[j]);
            o.x += p.beta * c0.x;
            o.y += p.beta * c0.y;
            o.z += p.beta * c0.z;
            o.w += p.beta * c0.w;
          }
          if (mok && cok[j]) *(float4*)(c32 + ncl[j]) = o;
        }
      }
    }
  } else {
    // row operand: residual (EPI_BF16 with HAS_OPT) or saved pre-activation (EPI_DGELU); HAS_OPT for EPI_GELU = write u
    constexpr bool HAS_OPND = (EPI == EPI_DGELU) || (EPI == EPI_BF16 && HAS_OPT);
    const bf16_t* opnd_p = (EPI == EPI_DGELU) ? p.aux_in : p.res;
    const int64_t opnd_ld = (EPI == EPI_DGELU) ? p.ldaux : p.ldr;
    u32x2_t opnd[2][FN];
    auto load_row = [&](int i, u32x2_t* dst) {
      int64_t m = m_base + i * 16 + frow;
      if constexpr (EDGE) m = m < p.M ? m : p.M - 1;
      const bf16_t* base = opnd_p + m * opnd_ld;
#pragma unroll
      for (int j = 0; j < FN; j++) dst[j] = *(const u32x2_t*)(base + ncl[j]);
    };
    if constexpr (HAS_OPND) load_row(0, opnd[0]);
    f32x2_t lrs[2];   // LNF: {rstd, -mean * rstd} of this lane's row, fetched one row block ahead like the row operands
    auto load_rs = [&](int i) {
      int64_t m = m_base + i * 16 + frow;
      if constexpr (EDGE) m = m < p.M ? m : p.M - 1;
      return *(const f32x2_t*)(p.lnf_rs + 2 * m);
    };
    if constexpr (LNF) lrs[0] = load_rs(0);
#pragma unroll
    for (int i = 0; i < FM; i++) {
      if constexpr (HAS_OPND) {
        if (i + 1 < FM) load_row(i + 1, opnd[(i + 1) & 1]);
      }
      if constexpr (LNF) {
        if (i + 1 < FM) lrs[(i + 1) & 1] = load_rs(i + 1);
      }
      const int64_t m = m_base + i * 16 + frow;
      const bool mok = EDGE ? m < p.M : true;
      const int64_t mc = mok ? m : p.M - 1;
      bf16_t* c16 = (bf16_t*)p.C + mc * p.ldc;
      bf16_t* auxo = (EPI == EPI_GELU && HAS_OPT) ? p.aux_out + mc * p.ldaux : nullptr;
#pragma unroll
      for (int j = 0; j < FN; j++) {
        f32x2_t v01, v23;
        if constexpr (LNF) {   // rstd * acc + (-mean * rstd) * c[n] + b'[n]
          const f32x2_t r2 = {lrs[i & 1][0], lrs[i & 1][0]}, s2 = {lrs[i & 1][1], lrs[i & 1][1]};
          v01 = __builtin_elementwise_fma((f32x2_t){acc[i][j][0], acc[i][j][1]}, r2,
                                          __builtin_elementwise_fma(s2, (f32x2_t){lc4[j].x, lc4[j].y}, (f32x2_t){bias4[j].x, bias4[j].y}));
          v23 = __builtin_elementwise_fma((f32x2_t){acc[i][j][2], acc[i][j][3]}, r2,
                                          __builtin_elementwise_fma(s2, (f32x2_t){lc4[j].z, lc4[j].w}, (f32x2_t){bias4[j].z, bias4[j].w}));
        } else {
          v01 = (f32x2_t){acc[i][j][0], acc[i][j][1]} + (f32x2_t){bias4[j].x, bias4[j].y};   // v_pk_add_f32
          v23 = (f32x2_t){acc[i][j][2], acc[i][j][3]} + (f32x2_t){bias4[j].z, bias4[j].w};
        }
        if constexpr (QS) {
          v01 *= qs2[j];
          v23 *= qs2[j];
        }
        if constexpr (EPI == EPI_GELU) {
          u32x2_t u;
          u[0] = pack_bf2_opaque(v01[0], v01[1]);
          u[1] = pack_bf2_opaque(v23[0], v23[1]);
          // GELU (and, for a layer that will run backward, gelu') of the bf16-ROUNDED pre-activation
          if constexpr (HAS_OPT) {
            f32x2_t d01, d23;
            if constexpr (LP) {
              gelu_dgelu2_lp((f32x2_t){bf_lo(u[0]), bf_hi(u[0])}, v01, d01);
              gelu_dgelu2_lp((f32x2_t){bf_lo(u[1]), bf_hi(u[1])}, v23, d23);
            } else {
              gelu_dgelu2((f32x2_t){bf_lo(u[0]), bf_hi(u[0])}, v01, d01);
              gelu_dgelu2((f32x2_t){bf_lo(u[1]), bf_hi(u[1])}, v23, d23);
            }
            u32x2_t dw;
            dw[0] = pack_bf2(d01[0], d01[1]);
            dw[1] = pack_bf2(d23[0], d23[1]);
            if (mok && cok[j]) *(u32x2_t*)(auxo + ncl[j]) = dw;
          } else if constexpr (LP) {
            v01 = gelu2_lp((f32x2_t){bf_lo(u[0]), bf_hi(u[0])});
            v23 = gelu2_lp((f32x2_t){bf_lo(u[1]), bf_hi(u[1])});
          } else {
            v01 = gelu2((f32x2_t){bf_lo(u[0]), bf_hi(u[0])});
            v23 = gelu2((f32x2_t){bf_lo(u[1]), bf_hi(u[1])});
          }
        } else if constexpr (EPI == EPI_DGELU) {
          const u32x2_t u = opnd[i & 1][j];   // saved gelu'(u)
          v01 *= (f32x2_t){bf_lo(u[0]), bf_hi(u[0])};
          v23 *= (f32x2_t){bf_lo(u[1]), bf_hi(u[1])};
        } else if constexpr (HAS_OPT) {
          const u32x2_t r2 = opnd[i & 1][j];
          v01 += (f32x2_t){bf_lo(r2[0]), bf_hi(r2[0])};
          v23 += (f32x2_t){bf_lo(r2[1]), bf_hi(r2[1])};
        }
        u32x2_t o;
        o[0] = pack_bf2(v01[0], v01[1]);
        o[1] = pack_bf2(v23[0], v23[1]);
        if (mok && cok[j]) *(u32x2_t*)(c16 + ncl[j]) = o;
      }
    }
  }
}

// bf16 epilogues of the 128 x 64 wave tile (8 x 4 MFMA blocks) staged through LDS.  In the MFMA C layout a store
// instruction covers 16 rows x 32 bytes: sixteen partial-line writes, and the L1 write path spends ~4 clocks on each
// (tools/gemm_ksweep.py: 7-9 us per 256 x 256 tile, ~25 % of a K = 1024 tile).  Here every wave converts its tile to
// bf16, parks it in a private 16 KB LDS region (the operand ring is dead by now; 8-byte writes, 16-byte chunks XOR-
// swizzled by (row >> 1) & 7: conflict-free reads, 2-way writes that hide under the ds_write data transfer) and reads
// it back row-major: one ds_read_b128 + one 16-byte store per lane, eight complete 128-byte lines per instruction.
// The arithmetic (bias, residual, GELU) stays in the MFMA layout and is identical to gemm_epilogue_impl.
// CSUM (EPI_DGELU only): the wave also sums its 128 x 64 output tile over the rows (fp32 values before the bf16 rounding;
// rows below `row_lo` -- the part of a SHIFTED edge tile that belongs to its neighbour -- are left out) and writes the 64
// column sums to p.colpart[slot][n_base ..]: du = dY of fc1 is produced here, so fc1's bias gradient costs 64 packed FMAs + 64
// DPP adds per wave tile instead of a second pass over du (colsum_bf16_kernel: 84 MB per ViT-L context block).
// PRE (round 5, persistent kernel only): 0 = straight passes -- the row operand (residual / saved gelu') of block i + 1 is requested while
// block i is computed, every pass is {eight staging writes, four times {read back, wait, store}}; 4 (default) = the row operand as
// sixteen row-major 16-byte loads issued before anything else (eight full 128-byte lines per instruction, parked in the staging area pass
// by pass and read back in the MFMA layout -- the mirror image of the output path) and software-PIPELINED passes (see the pass loop).
// Same values, same arithmetic: bit-identical outputs (profiles/r05_epi_pipeline.md; the intermediate forms 1 - 3 and the two
// diagnostic copies of round 5 are recorded there).
// NB: the launch has no bias (workgroup-uniform, resolved by the caller -- every dgrad GEMM): no bias registers (the dGELU epilogue with all its
// row operand in flight (PRE) and sixteen column-sum accumulators is otherwise 4 VGPRs over the budget, and hipcc's spill lands between the
// operand loads behind a vmcnt(0)) and no `+ bias` instruction (64 of a plain epilogue's ~200 vector instructions per wave tile).  Dropping
// `+ 0.0f` keeps the bits: an accumulator that starts at +0 and is only ever added to cannot hold -0 (x + (-x) and (+0) + (-0) are +0 in
// round-to-nearest), so there is no -0 for `+ 0.0f` to turn into +0.
template <int EPI, bool HAS_OPT, bool EDGE, int IPP, bool CSUM = false, bool QS = false, bool LP = false, bool LNF = false, int PRE = 0, bool NB = false>
__device__ __forceinline__ void gemm_epilogue_staged(const GemmArgs& p, f32x4_t (&acc)[8][4], int64_t m_base,
                                                     int64_t n_base, int frow, int fg, int lane, char* stage,
                                                     int64_t row_lo = 0, int slot = 0) {
  // IPP = 16-row blocks per pass: 8 -> the whole wave tile in one 16 KB pass (stage = 16 KB per wave, the dead operand
  // ring of the one-tile-per-workgroup kernel); 2 -> four 4 KB passes (persistent kernel: the ring already holds the
  // next tile's first parts, the staging area is a separate 32 KB).
  static_assert(EPI != EPI_F32, "fp32 outputs are stored directly");
  static_assert(IPP == 8 || IPP == 2, "passes of 128 or 32 rows");
  constexpr int FM = 8, FN = 4;
  const int64_t ncol0 = n_base + fg * 4;
  int64_t ncl[FN];
#pragma unroll
  for (int j = 0; j < FN; j++) {
    const int64_t n = ncol0 + j * 16;
    ncl[j] = (!EDGE || n < p.N) ? n : 0;
  }
  float4 bias4[NB ? 1 : FN];
  if constexpr (!NB) {
#pragma unroll
    for (int j = 0; j < FN; j++) bias4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) {
#pragma unroll
      for (int j = 0; j < FN; j++) bias4[j] = *(const float4*)(p.bias + ncl[j]);
    }
  }
  static_assert(!NB || !LNF, "no-bias variant: plain epilogues");
  static_assert(!LNF || ((EPI == EPI_BF16 || EPI == EPI_GELU) && !HAS_OPT && !CSUM), "LayerNorm fold: bf16 / GELU epilogue, no residual, no saved derivative");
  float4 lc4[LNF ? FN : 1];
  if constexpr (LNF) {
#pragma unroll
    for (int j = 0; j < FN; j++) lc4[j] = *(const float4*)(p.lnf_c + ncl[j]);
  }
  // LNF: {rstd, -mean * rstd} of this lane's eight rows, ALL requested here, before the first store of the epilogue: a load issued
  // between the stores would make its wait drain every store older than it (vmcnt counts stores), four times per tile
  f32x2_t lrs[LNF ? 8 : 1];
  if constexpr (LNF) {
#pragma unroll
    for (int i = 0; i < 8; i++) {
      int64_t m = m_base + i * 16 + frow;
      if constexpr (EDGE) m = m < p.M ? m : p.M - 1;
      lrs[i] = *(const f32x2_t*)(p.lnf_rs + 2 * m);
    }
  }
  constexpr bool HAS_OPND = (EPI == EPI_DGELU) || (EPI == EPI_BF16 && HAS_OPT);
  constexpr bool PIPE = PRE >= 4;                 // pipelined passes (also for epilogues without a row operand)
  constexpr bool PRM = HAS_OPND && PIPE;          // row operand as row-major pieces through the staging area
  static_assert(PRE == 0 || PRE == 4, "epilogue forms: straight (0) or pipelined (4)");
  static_assert(PRE == 0 || (IPP == 2 && !EDGE), "pipelined passes: interior tiles of the persistent kernel");
  const bf16_t* opnd_p = (EPI == EPI_DGELU) ? p.aux_in : p.res;
  const int64_t opnd_ld = (EPI == EPI_DGELU) ? p.ldaux : p.ldr;
  const int rrow = lane >> 3, rch = lane & 7;   // row-major side: 8 lanes per 128-byte row, 8 rows per instruction
  u32x4_t opnd_rm[PRM ? 2 * FM : 1];            // the row operand as sixteen row-major 16-byte pieces (rows it * 8 + rrow, chunk rch)
  if constexpr (PRM) {   // wave-uniform base (SGPRs, scalar arithmetic per row group) + one 32-bit lane offset: no vector address arithmetic
    // (the row-group stride passes through an opaque asm: hipcc otherwise hoists the sixteen products it * stride out of the TILE loop
    //  into SGPRs it then has to spill to vector lanes across the K loop)
    int64_t ostep = opnd_ld * 16;   // 8 rows, bytes
    asm volatile("" : "+s"(ostep));
    const char* ob = (const char*)(opnd_p + m_base * opnd_ld + n_base);
    const unsigned ooff = (unsigned)(rrow * (int)opnd_ld + rch * 8) * 2u;
#pragma unroll
    for (int it = 0; it < 2 * FM; it++) {
      opnd_rm[it] = *(const u32x4_t*)(ob + ooff);
      ob += ostep;
    }
  }
  // with a row operand in flight there is no blanket wait: the compiler's own counted waits let pass ps start when ITS four operand loads (and
  // the bias, and -- loads return in order -- every LDS-DMA issued before them) have landed, while the later passes' loads are still in flight
  if constexpr (!PRM) __builtin_amdgcn_s_waitcnt(0x0f70);   // vmcnt(0), compiler-visible (see gemm_epilogue_impl)

  // LDS addresses: write (MFMA layout) and read-back (row-major) sides of the same swizzled image
  int wr_off[FN];
  {
    const int key = (frow >> 1) & 7;
#pragma unroll
    for (int j = 0; j < FN; j++) wr_off[j] = frow * 128 + (((j * 2 + (fg >> 1)) ^ key) << 4) + (fg & 1) * 8;
  }
  int rd_off[2];   // read-back (rrow, rch above): 8 lanes per 128-byte row, 8 rows per instruction
#pragma unroll
  for (int par = 0; par < 2; par++) rd_off[par] = rrow * 128 + ((rch ^ ((par * 4 + (rrow >> 1)) & 7)) << 4);
  const bool col_ok = !EDGE || (n_base + rch * 8 < p.N);

  constexpr bool TWO_OUT = (EPI == EPI_GELU) && HAS_OPT;
  // 16-row blocks per pass.  Two outputs (GELU + saved derivative): half-size passes with BOTH images in the stage at once
  // (blocks 0 .. RPP-1: the derivative, RPP .. 2 RPP-1: the GELU output) -- holding the second output in registers until the
  // first is flushed costs 16 ... 64 VGPRs next to 128 live accumulators and made hipcc spill inside the K loop.
  constexpr int RPP = TWO_OUT ? IPP / 2 : IPP;
  static_assert(RPP >= 1 && FM % RPP == 0, "pass size");

  // LDS image (RPP blocks starting at stage block `blk0`) -> global: (ds_read_b128 + 16-byte store) per 8 rows
  auto flush = [&](bf16_t* out, int64_t ld, int row0, int blk0) {
#pragma unroll
    for (int it = 0; it < RPP * 2; it++) {
      const u32x4_t v = *(const u32x4_t*)(stage + blk0 * 2048 + it * 1024 + rd_off[it & 1]);
      const int64_t m = m_base + row0 + it * 8 + rrow;
#ifdef VJ_NT_STORE   // A/B build only (python -m jepa_amd.build nts -DVJ_NT_STORE=1): the outputs bypass L2 retention
      if (col_ok && (!EDGE || m < p.M)) __builtin_nontemporal_store(v, (u32x4_t*)(out + m * ld + n_base + rch * 8));
#else
      if (col_ok && (!EDGE || m < p.M)) *(u32x4_t*)(out + m * ld + n_base + rch * 8) = v;
#endif
    }
  };

  u32x2_t opnd[2][FN];
  auto load_row = [&](int i, u32x2_t* dst) {
    int64_t m = m_base + i * 16 + frow;
    if constexpr (EDGE) m = m < p.M ? m : p.M - 1;
    const bf16_t* base = opnd_p + m * opnd_ld;
#pragma unroll
    for (int j = 0; j < FN; j++) dst[j] = *(const u32x2_t*)(base + ncl[j]);
  };
  static_assert(!CSUM || (EPI == EPI_DGELU && !EDGE), "column sums: the fc2-dgrad epilogue of the persistent kernel");
  static_assert(!QS || (EPI == EPI_BF16 && !HAS_OPT), "column scale: the qkv projection (bf16 output, no residual)");
  f32x2_t qs2[QS ? FN : 1];
  if constexpr (QS) {
#pragma unroll
    for (int j = 0; j < FN; j++) {
      const float sj = (ncol0 + j * 16 < p.qcols) ? p.qscale : 1.0f;
      qs2[j] = (f32x2_t){sj, sj};
    }
  }
  f32x2_t cs01[CSUM ? FN : 1], cs23[CSUM ? FN : 1];
  if constexpr (CSUM) {
#pragma unroll
    for (int j = 0; j < FN; j++) cs01[j] = cs23[j] = (f32x2_t){0.f, 0.f};
  }
  const int row_first = (int)(row_lo - m_base) - frow;   // CSUM: block i of this lane counts iff i * 16 >= row_first

  // one element group (block i, column tile j): accumulator -> the bf16 output words `o` (and the saved derivative `dw`); `u` = this lane's
  // 8 bytes of the row operand (residual / saved gelu'), whatever way it reached the lane
  auto elem = [&](int i, int j, f32x2_t mk2, u32x2_t u, u32x2_t& o, u32x2_t& dw) __attribute__((always_inline)) {
    f32x2_t v01, v23;
    if constexpr (LNF) {   // rstd * acc + (-mean * rstd) * c[n] + b'[n]
      const f32x2_t r2 = {lrs[i][0], lrs[i][0]}, s2 = {lrs[i][1], lrs[i][1]};
      v01 = __builtin_elementwise_fma((f32x2_t){acc[i][j][0], acc[i][j][1]}, r2,
                                      __builtin_elementwise_fma(s2, (f32x2_t){lc4[j].x, lc4[j].y}, (f32x2_t){bias4[j].x, bias4[j].y}));
      v23 = __builtin_elementwise_fma((f32x2_t){acc[i][j][2], acc[i][j][3]}, r2,
                                      __builtin_elementwise_fma(s2, (f32x2_t){lc4[j].z, lc4[j].w}, (f32x2_t){bias4[j].z, bias4[j].w}));
    } else if constexpr (NB) {
      v01 = (f32x2_t){acc[i][j][0], acc[i][j][1]};
      v23 = (f32x2_t){acc[i][j][2], acc[i][j][3]};
    } else {
      v01 = (f32x2_t){acc[i][j][0], acc[i][j][1]} + (f32x2_t){bias4[j].x, bias4[j].y};   // v_pk_add_f32
      v23 = (f32x2_t){acc[i][j][2], acc[i][j][3]} + (f32x2_t){bias4[j].z, bias4[j].w};
    }
    if constexpr (QS) {
      v01 *= qs2[j];
      v23 *= qs2[j];
    }
    if constexpr (EPI == EPI_GELU) {
      u32x2_t w;
      w[0] = pack_bf2_opaque(v01[0], v01[1]);
      w[1] = pack_bf2_opaque(v23[0], v23[1]);
      // GELU (and, for a layer that will run backward, gelu') of the bf16-ROUNDED pre-activation
      if constexpr (TWO_OUT) {
        f32x2_t d01, d23;
        if constexpr (LP) {
          gelu_dgelu2_lp((f32x2_t){bf_lo(w[0]), bf_hi(w[0])}, v01, d01);
          gelu_dgelu2_lp((f32x2_t){bf_lo(w[1]), bf_hi(w[1])}, v23, d23);
        } else {
          gelu_dgelu2((f32x2_t){bf_lo(w[0]), bf_hi(w[0])}, v01, d01);
          gelu_dgelu2((f32x2_t){bf_lo(w[1]), bf_hi(w[1])}, v23, d23);
        }
        dw[0] = pack_bf2(d01[0], d01[1]);
        dw[1] = pack_bf2(d23[0], d23[1]);
      } else if constexpr (LP) {
        v01 = gelu2_lp((f32x2_t){bf_lo(w[0]), bf_hi(w[0])});
        v23 = gelu2_lp((f32x2_t){bf_lo(w[1]), bf_hi(w[1])});
      } else {
        v01 = gelu2((f32x2_t){bf_lo(w[0]), bf_hi(w[0])});
        v23 = gelu2((f32x2_t){bf_lo(w[1]), bf_hi(w[1])});
      }
    } else if constexpr (EPI == EPI_DGELU) {
      v01 *= (f32x2_t){bf_lo(u[0]), bf_hi(u[0])};   // u = saved gelu'(u)
      v23 *= (f32x2_t){bf_lo(u[1]), bf_hi(u[1])};
      if constexpr (CSUM) {
        cs01[j] = __builtin_elementwise_fma(v01, mk2, cs01[j]);
        cs23[j] = __builtin_elementwise_fma(v23, mk2, cs23[j]);
      }
    } else if constexpr (HAS_OPT) {
      v01 += (f32x2_t){bf_lo(u[0]), bf_hi(u[0])};   // u = residual
      v23 += (f32x2_t){bf_lo(u[1]), bf_hi(u[1])};
    }
    o[0] = pack_bf2(v01[0], v01[1]);
    o[1] = pack_bf2(v23[0], v23[1]);
  };
  auto row_mask = [&](int i) __attribute__((always_inline)) {
    f32x2_t mk2 = {1.f, 1.f};
    if constexpr (CSUM) {
      const float mk = (i * 16 >= row_first) ? 1.f : 0.f;
      mk2 = (f32x2_t){mk, mk};
    }
    return mk2;
  };

  if constexpr (!PIPE) {
    if constexpr (HAS_OPND) load_row(0, opnd[0]);
#pragma unroll
    for (int ps = 0; ps < FM / RPP; ps++) {
#pragma unroll
      for (int ii = 0; ii < RPP; ii++) {
        const int i = ps * RPP + ii;
        if constexpr (HAS_OPND) {
          if (i + 1 < FM) load_row(i + 1, opnd[(i + 1) & 1]);
        }
        const f32x2_t mk2 = row_mask(i);
#pragma unroll
        for (int j = 0; j < FN; j++) {
          u32x2_t u = {0u, 0u}, o, dw;
          if constexpr (HAS_OPND) u = opnd[i & 1][j];
          elem(i, j, mk2, u, o, dw);
          if constexpr (TWO_OUT) *(u32x2_t*)(stage + ii * 2048 + wr_off[j]) = dw;   // the saved derivative: stage blocks 0 .. RPP-1
          *(u32x2_t*)(stage + ((TWO_OUT ? RPP : 0) + ii) * 2048 + wr_off[j]) = o;
        }
      }
      if constexpr (TWO_OUT) flush(p.aux_out, p.ldaux, ps * RPP * 16, 0);
      flush((bf16_t*)p.C, p.ldc, ps * RPP * 16, TWO_OUT ? RPP : 0);
    }
  } else {
    // PIPELINED passes (PRE 4).  The straight form above is, per wave, a chain of exposed LDS round trips: eight writes, then four times
    // {ds_read_b128, s_waitcnt lgkmcnt(0), global_store} (hipcc keeps the source order), then the next pass's arithmetic -- sixteen serialised
    // round trips per wave tile on an LDS that all eight waves use at once; the phase stamps (tools/gemm_stamps.py) put 2.9 ... 3.5 us of a
    // 25 us K = 1024 tile into an epilogue whose LDS, store and vector work are ~1 us each.  Here a pass is
    //     write O(ps) -> issue the four row-major reads V(ps) -> [park the operand rows of pass ps + 1, read them back in the MFMA layout]
    //     -> the arithmetic of pass ps + 1 into registers O(ps + 1), under the reads' latency -> store V(ps)
    // (LDS operations of a wave execute in order, so overwriting the image behind the issued reads is safe.)  Same values, same arithmetic:
    // bit-identical outputs.
    constexpr int NP = FM / RPP;
    constexpr int NV = TWO_OUT ? 4 * RPP : 2 * RPP;
    u32x2_t O[RPP][FN], Dw[TWO_OUT ? RPP : 1][FN];
    auto park = [&](int ps) __attribute__((always_inline)) {
#pragma unroll
      for (int it = 0; it < RPP * 2; it++) *(u32x4_t*)(stage + it * 1024 + rd_off[it & 1]) = opnd_rm[ps * RPP * 2 + it];
    };
    auto compute = [&](int ps) __attribute__((always_inline)) {
      u32x2_t U[RPP][FN];
#pragma unroll
      for (int ii = 0; ii < RPP; ii++)
#pragma unroll
        for (int j = 0; j < FN; j++) {
          if constexpr (PRM) U[ii][j] = *(const u32x2_t*)(stage + ii * 2048 + wr_off[j]);
          else U[ii][j] = (u32x2_t){0u, 0u};
        }
#pragma unroll
      for (int ii = 0; ii < RPP; ii++) {
        const int i = ps * RPP + ii;
        const f32x2_t mk2 = row_mask(i);
#pragma unroll
        for (int j = 0; j < FN; j++) {
          u32x2_t dw;
          elem(i, j, mk2, U[ii][j], O[ii][j], dw);
          if constexpr (TWO_OUT) Dw[ii][j] = dw;
        }
      }
    };
    // pass 0 in the straight form (element by element into the stage: with all 128 accumulators and the whole row operand still live there
    // are no registers for a pass of outputs)
    if constexpr (PRM) park(0);
#pragma unroll
    for (int ii = 0; ii < RPP; ii++) {
      const f32x2_t mk2 = row_mask(ii);
#pragma unroll
      for (int j = 0; j < FN; j++) {
        u32x2_t u = {0u, 0u}, o, dw;
        if constexpr (PRM) u = *(const u32x2_t*)(stage + ii * 2048 + wr_off[j]);
        elem(ii, j, mk2, u, o, dw);
        if constexpr (TWO_OUT) *(u32x2_t*)(stage + ii * 2048 + wr_off[j]) = dw;
        *(u32x2_t*)(stage + ((TWO_OUT ? RPP : 0) + ii) * 2048 + wr_off[j]) = o;
      }
    }
    // output row-group pointers / strides (the strides pass through an opaque asm: hipcc otherwise turns the running pointer back into
    // sixteen products it * stride, hoists them out of the TILE loop and spills the SGPRs to vector lanes across the K loop)
    int64_t step_c = p.ldc * 16, step_aux = TWO_OUT ? p.ldaux * 16 : 0;   // 8 rows, bytes
    asm volatile("" : "+s"(step_c));
    if constexpr (TWO_OUT) asm volatile("" : "+s"(step_aux));
    char* ob_c = (char*)((bf16_t*)p.C + m_base * p.ldc + n_base);
    char* ob_aux = TWO_OUT ? (char*)(p.aux_out + m_base * p.ldaux + n_base) : nullptr;
    const unsigned ooff_c = (unsigned)(rrow * (int)p.ldc + rch * 8) * 2u;
    const unsigned ooff_aux = TWO_OUT ? (unsigned)(rrow * (int)p.ldaux + rch * 8) * 2u : 0u;
    auto read_back = [&](u32x4_t (&V)[NV]) __attribute__((always_inline)) {
#pragma unroll
      for (int it = 0; it < NV; it++) {
        V[it] = *(const u32x4_t*)(stage + it * 1024 + rd_off[it & 1]);   // (two outputs: the derivative's blocks first)
      }
    };
    auto store_out = [&](const u32x4_t (&V)[NV]) __attribute__((always_inline)) {
#pragma unroll
      for (int it = 0; it < NV; it++) {
        const bool second = TWO_OUT && it >= 2 * RPP;                  // two outputs: V[0 .. 2 RPP) -> aux_out, the rest -> C
        // wave-uniform row-group pointer (advanced by scalar adds) + one 32-bit lane offset: global_store ... s[base] form.  The vector
        // form cost 47 of the ~300 instructions of a plain epilogue, whose time is the issue of its instructions (tools/gemm_stamps.py)
        char*& ob = (TWO_OUT && !second) ? ob_aux : ob_c;
        const unsigned ooff = (TWO_OUT && !second) ? ooff_aux : ooff_c;
        *(u32x4_t*)(ob + ooff) = V[it];
        ob += (TWO_OUT && !second) ? step_aux : step_c;
      }
    };
    auto put = [&]() __attribute__((always_inline)) {   // O (and Dw) -> the stage image
#pragma unroll
      for (int ii = 0; ii < RPP; ii++)
#pragma unroll
        for (int j = 0; j < FN; j++) {
          if constexpr (TWO_OUT) *(u32x2_t*)(stage + ii * 2048 + wr_off[j]) = Dw[ii][j];
          *(u32x2_t*)(stage + ((TWO_OUT ? RPP : 0) + ii) * 2048 + wr_off[j]) = O[ii][j];
        }
    };
#pragma unroll
    for (int ps = 0; ps < NP; ps++) {
      u32x4_t V[NV];
      read_back(V);
      __builtin_amdgcn_sched_barrier(0);
      if (ps + 1 < NP) {
        if constexpr (PRM) park(ps + 1);
        compute(ps + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      store_out(V);
      if (ps + 1 < NP) put();
    }
    // (one pass deeper -- the image of pass ps + 1 written and its read-back issued before the stores of pass ps -- was built and measured
    //  level: profiles/r05_epi_pipeline.md)
  }
  if constexpr (CSUM) {
    // sum over the 16 rows (lanes frow = 0..15 of each 16-lane DPP row hold the same columns): rotate-and-add, fixed order
    float cv[FN][4];
#pragma unroll
    for (int j = 0; j < FN; j++) {
      cv[j][0] = cs01[j][0];
      cv[j][1] = cs01[j][1];
      cv[j][2] = cs23[j][0];
      cv[j][3] = cs23[j][1];
#pragma unroll
      for (int c = 0; c < 4; c++) cv[j][c] = row16_sum(cv[j][c]);
    }
    // (lane id re-derived through a volatile asm: the per-lane address below is then computed HERE, not hoisted to the top of the epilogue,
    //  where it would be one 64-bit value too many next to 128 accumulators + 64 operand registers + 16 sums -- hipcc spilled it, and the
    //  reload's wait behind the epilogue's stores drained them)
    int l2;
    asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l2));
    if ((l2 & 15) == 0) {
      float* cp = p.colpart + (int64_t)slot * p.N + n_base + (l2 >> 4) * 4;
#pragma unroll
      for (int j = 0; j < FN; j++) *(float4*)(cp + j * 16) = make_float4(cv[j][0], cv[j][1], cv[j][2], cv[j][3]);
    }
  }
}

// staged variant selector for the 8-phase kernel (wave tile 128 x 64); falls back to the direct form when the 16-byte
// row-major stores cannot be used (N, ldc or the base pointers not 8-element aligned) and for fp32 outputs
// (ALLOW_LNF = false: kernels that are never launched with a folded LayerNorm leave those variants out -- the 4-wave kernels'
//  scalar register budget is spent on their twelve wave-uniform DMA bases)
template <int EPI, int IPP = 8, bool ALLOW_LNF = true, int PRE = 0>
__device__ __forceinline__ bool gemm_epilogue_try_staged(const GemmArgs& p, f32x4_t (&acc)[8][4], int64_t m_base,
                                                         int64_t n_base, int frow, int fg, int lane, char* stage,
                                                         int64_t row_lo = 0, int slot = 0) {
  if constexpr (EPI == EPI_F32) {
    return false;
  } else {
    bool ok = (p.N % 8 == 0) && (p.ldc % 8 == 0) && (((uintptr_t)p.C & 15) == 0);
    if constexpr (EPI == EPI_GELU) ok = ok && (p.aux_out == nullptr || ((p.ldaux % 8 == 0) && (((uintptr_t)p.aux_out & 15) == 0)));
    if (!ok) return false;
    bool opt;
    if constexpr (EPI == EPI_GELU) opt = p.aux_out != nullptr;
    else if constexpr (EPI == EPI_BF16) opt = p.res != nullptr;
    else opt = true;
    const bool edge = __builtin_amdgcn_readfirstlane((m_base + 128 > p.M) || (n_base + 64 > p.N));
    constexpr int XP = (PRE >= 4 && IPP == 2) ? PRE : 0;   // pipelined passes: every interior variant of the persistent kernel
    if constexpr (EPI == EPI_DGELU && IPP == 2) {   // persistent kernel (every tile interior): optional fused column sums
      if (p.colpart != nullptr && !edge) {
        if constexpr (PRE != 0) {   // (a dgrad GEMM has no bias; with one, the kernel that keeps bias registers)
          if (p.bias == nullptr) gemm_epilogue_staged<EPI, true, false, IPP, true, false, false, false, PRE, true>(p, acc, m_base, n_base, frow, fg, lane, stage, row_lo, slot);
          else gemm_epilogue_staged<EPI, true, false, IPP, true>(p, acc, m_base, n_base, frow, fg, lane, stage, row_lo, slot);
        } else {
          gemm_epilogue_staged<EPI, true, false, IPP, true>(p, acc, m_base, n_base, frow, fg, lane, stage, row_lo, slot);
        }
        return true;
      }
    }
    // row operand requested up front (option gemm_epi_pre -> a kernel of its own: the variant is a compile-time property, so that the
    // default kernels' register allocation is untouched)
    if constexpr (PRE != 0 && (EPI == EPI_DGELU || EPI == EPI_BF16) && IPP == 2) {
      if (opt && !edge && (EPI == EPI_DGELU ? p.bias == nullptr : p.lnf_rs == nullptr)) {
        if constexpr (EPI == EPI_BF16) {   // (pipelined form: a residual GEMM without a bias -- the dgrad that adds the skip path's gradient)
          if (p.bias == nullptr) {
            gemm_epilogue_staged<EPI, true, false, IPP, false, false, false, false, PRE, true>(p, acc, m_base, n_base, frow, fg, lane, stage);
            return true;
          }
        }
        gemm_epilogue_staged<EPI, true, false, IPP, false, false, false, false, PRE, EPI == EPI_DGELU>(p, acc, m_base, n_base, frow, fg, lane, stage);
        return true;
      }
    }
    if constexpr (ALLOW_LNF && (EPI == EPI_BF16 || EPI == EPI_GELU)) {
      if (p.lnf_rs != nullptr) {   // LayerNorm folded into this GEMM (workgroup-uniform; the launcher guarantees: no residual / aux_out)
        if constexpr (EPI == EPI_BF16) {
          if (p.qscale != 0.f && n_base < p.qcols) {
            if (edge) gemm_epilogue_staged<EPI, false, true, IPP, false, true, false, true>(p, acc, m_base, n_base, frow, fg, lane, stage);
            else gemm_epilogue_staged<EPI, false, false, IPP, false, true, false, true, XP>(p, acc, m_base, n_base, frow, fg, lane, stage);
          } else {
            if (edge) gemm_epilogue_staged<EPI, false, true, IPP, false, false, false, true>(p, acc, m_base, n_base, frow, fg, lane, stage);
            else gemm_epilogue_staged<EPI, false, false, IPP, false, false, false, true, XP>(p, acc, m_base, n_base, frow, fg, lane, stage);
          }
        } else {
          if (edge) gemm_epilogue_staged<EPI, false, true, IPP, false, false, true, true>(p, acc, m_base, n_base, frow, fg, lane, stage);
          else gemm_epilogue_staged<EPI, false, false, IPP, false, false, true, true, XP>(p, acc, m_base, n_base, frow, fg, lane, stage);
        }
        return true;
      }
    }
    if constexpr (EPI == EPI_BF16) {
      if (p.qscale != 0.f && n_base < p.qcols) {   // wave tiles that hold q columns only (the launcher guarantees: no residual)
        if (edge) gemm_epilogue_staged<EPI, false, true, IPP, false, true>(p, acc, m_base, n_base, frow, fg, lane, stage);
        else gemm_epilogue_staged<EPI, false, false, IPP, false, true, false, false, XP>(p, acc, m_base, n_base, frow, fg, lane, stage);
        return true;
      }
    }
    if constexpr (EPI == EPI_GELU) {   // GELU: Phi(-|x|) as exp2 of a degree-6 polynomial (LP = true; the Abramowitz-Stegun form of rounds 1-3 is gone)
      if (opt) {
        if (edge) gemm_epilogue_staged<EPI, true, true, IPP, false, false, true>(p, acc, m_base, n_base, frow, fg, lane, stage);
        else gemm_epilogue_staged<EPI, true, false, IPP, false, false, true, false, XP>(p, acc, m_base, n_base, frow, fg, lane, stage);
      } else {
        if (edge) gemm_epilogue_staged<EPI, false, true, IPP, false, false, true>(p, acc, m_base, n_base, frow, fg, lane, stage);
        else gemm_epilogue_staged<EPI, false, false, IPP, false, false, true, false, XP>(p, acc, m_base, n_base, frow, fg, lane, stage);
      }
      return true;
    } else if (opt) {
      if (edge) gemm_epilogue_staged<EPI, true, true, IPP>(p, acc, m_base, n_base, frow, fg, lane, stage);
      else gemm_epilogue_staged<EPI, true, false, IPP, false, false, false, false, EPI == EPI_DGELU ? 0 : XP>(p, acc, m_base, n_base, frow, fg, lane, stage);   // (dGELU WITH a bias: the straight form)
    } else if constexpr (EPI != EPI_DGELU) {
      if (edge) {
        gemm_epilogue_staged<EPI, false, true, IPP>(p, acc, m_base, n_base, frow, fg, lane, stage);
      } else {
        if constexpr (EPI == EPI_BF16 && XP >= 4) {   // (pipelined form: plain dgrad GEMMs have no bias)
          if (p.bias == nullptr) {
            gemm_epilogue_staged<EPI, false, false, IPP, false, false, false, false, XP, true>(p, acc, m_base, n_base, frow, fg, lane, stage);
            return true;
          }
        }
        gemm_epilogue_staged<EPI, false, false, IPP, false, false, false, false, XP>(p, acc, m_base, n_base, frow, fg, lane, stage);
      }
    }
    return true;
  }
}

template <int EPI, int FM, int FN, bool INTERIOR_VARIANT = true, bool ALLOW_LNF = true>
__device__ __forceinline__ void gemm_epilogue(const GemmArgs& p, f32x4_t (&acc)[FM][FN], int64_t m_base,
                                              int64_t n_base, int frow, int fg, int slice) {
  bool opt;   // workgroup-uniform: resolved once, each variant is branch-free inside
  if constexpr (EPI == EPI_F32) opt = p.splitk > 1;
  else if constexpr (EPI == EPI_GELU) opt = p.aux_out != nullptr;
  else if constexpr (EPI == EPI_BF16) opt = p.res != nullptr;
  else opt = true;
  // interior wave tiles (the vast majority) carry no predicates at all
  const bool edge =
      !INTERIOR_VARIANT || __builtin_amdgcn_readfirstlane((m_base + FM * 16 > p.M) || (n_base + FN * 16 > p.N));
  if constexpr (ALLOW_LNF && (EPI == EPI_BF16 || EPI == EPI_GELU)) {
    if (p.lnf_rs != nullptr) {   // LayerNorm folded into this GEMM (the launcher guarantees: no residual / aux_out); always the predicated form
      if constexpr (EPI == EPI_BF16) {
        if (p.qscale != 0.f && n_base < p.qcols) gemm_epilogue_impl<EPI, FM, FN, false, true, false, true, false, true>(p, acc, m_base, n_base, frow, fg, slice);
        else gemm_epilogue_impl<EPI, FM, FN, false, true, false, false, false, true>(p, acc, m_base, n_base, frow, fg, slice);
      } else {
        gemm_epilogue_impl<EPI, FM, FN, false, true, false, false, true, true>(p, acc, m_base, n_base, frow, fg, slice);
      }
      return;
    }
  }
  if constexpr (EPI == EPI_BF16) {
    if (p.qscale != 0.f && n_base < p.qcols) {   // wave tiles that hold q columns only (the launcher guarantees: no residual)
      if (edge) gemm_epilogue_impl<EPI, FM, FN, false, true, false, true>(p, acc, m_base, n_base, frow, fg, slice);
      else gemm_epilogue_impl<EPI, FM, FN, false, false, false, true>(p, acc, m_base, n_base, frow, fg, slice);
      return;
    }
  }
  if constexpr (EPI == EPI_GELU) {   // (LP = true: the polynomial form, the only one since round 6)
    if (opt) {
      if (edge) gemm_epilogue_impl<EPI, FM, FN, true, true, false, false, true>(p, acc, m_base, n_base, frow, fg, slice);
      else gemm_epilogue_impl<EPI, FM, FN, true, false, false, false, true>(p, acc, m_base, n_base, frow, fg, slice);
    } else {
      if (edge) gemm_epilogue_impl<EPI, FM, FN, false, true, false, false, true>(p, acc, m_base, n_base, frow, fg, slice);
      else gemm_epilogue_impl<EPI, FM, FN, false, false, false, false, true>(p, acc, m_base, n_base, frow, fg, slice);
    }
    return;
  } else if (opt) {
    if (edge) gemm_epilogue_impl<EPI, FM, FN, true, true>(p, acc, m_base, n_base, frow, fg, slice);
    else gemm_epilogue_impl<EPI, FM, FN, true, false>(p, acc, m_base, n_base, frow, fg, slice);
  } else if constexpr (EPI == EPI_F32) {
    if (p.beta != 0.f) gemm_epilogue_impl<EPI, FM, FN, false, true, true>(p, acc, m_base, n_base, frow, fg, slice);
    else if (edge) gemm_epilogue_impl<EPI, FM, FN, false, true>(p, acc, m_base, n_base, frow, fg, slice);
    else gemm_epilogue_impl<EPI, FM, FN, false, false>(p, acc, m_base, n_base, frow, fg, slice);
  } else if constexpr (EPI != EPI_DGELU) {
    if (edge) gemm_epilogue_impl<EPI, FM, FN, false, true>(p, acc, m_base, n_base, frow, fg, slice);
    else gemm_epilogue_impl<EPI, FM, FN, false, false>(p, acc, m_base, n_base, frow, fg, slice);
  }
}

// Split-K factor for the fp32 weight-gradient GEMMs (few output tiles, very long K).  Cost model in units of one K-tile
// of one workgroup round: rounds x (K-tiles per slice + fixed per-workgroup overhead) + the HBM time of writing and
// re-reading the `s` fp32 partial matrices.  Picks whole rounds of `slots` concurrent workgroups instead of the
// "just above one round" counts a plain ceil() produces (48 tiles x 6 slices = 288 workgroups = two rounds at 56 %).
static inline int pick_splitk(int64_t tiles, int nk, int slots, double us_per_ktile, int min_ktiles, int64_t M,
                              int64_t N, int64_t ws_bytes) {
  int64_t smax = nk / min_ktiles > 0 ? nk / min_ktiles : 1;
  if (smax > 64) smax = 64;
  while (smax > 1 && smax * M * N * 4 > ws_bytes) smax--;
  const double partial_us = 2.0 * (double)M * (double)N * 4.0 / 3.0e6;   // write + read of one partial at ~3 TB/s
  int best = 1;
  double best_cost = 1e30;
  for (int64_t sk = 1; sk <= smax; sk++) {
    const int64_t per = (nk + sk - 1) / sk;
    const int64_t eff = (nk + per - 1) / per;   // no empty slices
    if (eff != sk) continue;
    const int64_t rounds = (tiles * sk + slots - 1) / slots;
    const double cost = (double)rounds * ((double)per + 4.0) * us_per_ktile + (sk > 1 ? (double)sk * partial_us : 0.0);
    if (cost < best_cost) {
      best_cost = cost;
      best = (int)sk;
    }
  }
  return best;
}

// XCD-aware, grouped tile mapping (bijective for any grid size): workgroup `bid` of `nblk` -> logical tile index such
// that each XCD (private L2; hardware dispatches workgroup b to XCD b % 8) works on a contiguous band of tiles.
__device__ __forceinline__ int xcd_logical(int bid, int nblk) {
  const int qx = nblk >> 3, rx = nblk & 7, xcd = bid & 7, pos = bid >> 3;
  return (xcd < rx ? xcd * (qx + 1) : rx * (qx + 1) + (xcd - rx) * qx) + pos;
}
// GM row-tiles form a group that sweeps all column tiles (keeps the A panel hot in L2)
// raster (persistent kernel only, option gemm_raster): bits 0-7 = group size G (0 -> 8); bit 8 clear = G ROW tiles sweep all column tiles,
// row tile fastest (the default: 32 consecutive tiles = 8 rows x 4 columns, the A panels are revisited column step after column step);
// bit 8 set = G COLUMN tiles sweep all row tiles, column tile fastest (32 consecutive tiles = 8 rows x 4 columns again, but an XCD's band
// now walks DOWN the rows of one column group: its B panels stay in the L2 while the A panels stream through once per column group)
__device__ __forceinline__ void tile_of_raster(int logical, int tiles_m, int tiles_n, int raster, int& tm, int& tn) {
  const int G = (raster & 0xff) ? (raster & 0xff) : 8;
  if (raster & 0x100) {
    const int per_group = G * tiles_m;
    const int group = logical / per_group, in_g = logical - group * per_group;
    const int first_n = group * G;
    const int gsz = (tiles_n - first_n) < G ? (tiles_n - first_n) : G;
    tn = first_n + in_g % gsz;
    tm = in_g / gsz;
  } else {
    const int per_group = G * tiles_n;
    const int group = logical / per_group, in_g = logical - group * per_group;
    const int first_m = group * G;
    const int gsz = (tiles_m - first_m) < G ? (tiles_m - first_m) : G;
    tm = first_m + in_g % gsz;
    tn = in_g / gsz;
  }
}
__device__ __forceinline__ void tile_of(int logical, int tiles_m, int tiles_n, int& tm, int& tn) {
  constexpr int GM = 8;
  const int per_group = GM * tiles_n;
  const int group = logical / per_group, in_g = logical - group * per_group;
  const int first_m = group * GM;
  const int gsz = (tiles_m - first_m) < GM ? (tiles_m - first_m) : GM;
  tm = first_m + in_g % gsz;
  tn = in_g / gsz;
}
