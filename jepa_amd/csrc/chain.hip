// Whole-trunk launch chains: N transformer blocks forward / backward enqueued by ONE C call.
//
// Reference semantics (restated, never copied): Block / Attention / MLP forward of src/models/utils/modules.py:30-36,
// 61-78,114-120 and what autograd derives from them for app/vjepa/train.py:461-464.  The Python engine used to issue the
// ~1800 kernel launches of a step one ctypes call at a time (~39 us each: 70 ms of host time against an 85 ms GPU step);
// here the per-block sequence lives behind the C ABI, so the host cost of a block is 7 (forward) / ~25 (backward)
// hipLaunchKernel calls and nothing else.  No arithmetic happens in this file: it only sequences the kernels of
// gemm*.hip / attention.hip / norm_loss.hip / rows.hip, lays the saved activations out in a caller-provided workspace
// and orders the two HIP streams (dgrad chain on `stream`, weight gradients on `side`) with events.
//
// Saved-activation layout per block (workspace `save_ws`, everything 256-byte aligned, bf16 unless noted):
//   x [M,D] (block input; block 0 uses the caller's x_in), y1 [M,D], qkv [M,3D], o [M,D], x1 [M,D], y2 [M,D],
//   u [M,Dh] (gelu'(pre-activation): the saved GELU derivative), g [M,Dh], mean1 rstd1 mean2 rstd2 [M] fp32, lse2 [H*M] fp32 (per segment [B,H,S]).
// With save = 0 (EMA target encoder, inference) one such set is reused by every block and x ping-pongs.
#include "common.hpp"
#include "options.hpp"
#include "../../include/vjepa_hip.h"
#include <atomic>
#include <mutex>
#include <vector>
#include <string>
#include <cstdlib>
#include <algorithm>

#define CH(call)              \
  do {                        \
    int _rc = (call);         \
    if (_rc != 0) return _rc; \
  } while (0)
#define HIPCH(call, what)                                                    \
  do {                                                                       \
    hipError_t _e = (call);                                                  \
    if (_e != hipSuccess) {                                                  \
      vj_set_error("%s: %s", what, hipGetErrorString(_e));                   \
      return (int)_e;                                                        \
    }                                                                        \
  } while (0)

static inline int64_t al256(int64_t n) { return (n + 255) / 256 * 256; }
static inline int64_t pad64i(int64_t m) { return (m + 63) / 64 * 64; }

// ---------------------------------------------------------------------------------------------------- guard bands
// Option ws_guard (diagnostics, tests/test_chain_gpu.py): every member of the two workspace layouts is followed by a 256-byte
// gap.  A chain call fills the gaps of the workspace it was given with a byte pattern (in stream order, before its first
// kernel) and remembers where they are; vj_ws_guard_check() synchronises the device and counts the gaps whose pattern
// changed -- a kernel that writes past the end of a saved activation, a column-partial or a split-K buffer lands in one.
#define GUARD_BYTES 256
#define GUARD_PATTERN 0xA5
static inline int64_t guard_gap() { return vj_opt(VJ_OPT_WS_GUARD) ? GUARD_BYTES : 0; }
namespace {
std::mutex g_guard_mu;
std::vector<char*> g_guards;   // device addresses of the gaps poisoned and not yet inspected
int64_t g_guard_checked = 0, g_guard_bad = 0;
char* g_guard_first_bad = nullptr;

// inspect (and forget) the recorded gaps inside [lo, hi); the device must be idle.  Caller holds g_guard_mu.
int inspect_gaps(char* lo, char* hi) {
  std::sort(g_guards.begin(), g_guards.end());
  g_guards.erase(std::unique(g_guards.begin(), g_guards.end()), g_guards.end());
  std::vector<char*> keep;
  unsigned char host[GUARD_BYTES];
  for (char* p : g_guards) {
    if (p < lo || p >= hi) {
      keep.push_back(p);
      continue;
    }
    hipError_t e = hipMemcpy(host, p, GUARD_BYTES, hipMemcpyDeviceToHost);
    if (e != hipSuccess) {
      vj_set_error("ws_guard: %s", hipGetErrorString(e));
      return (int)e;
    }
    bool bad = false;
    for (int i = 0; i < GUARD_BYTES; i++) bad |= host[i] != GUARD_PATTERN;
    g_guard_checked++;
    if (bad) {
      if (g_guard_bad == 0) g_guard_first_bad = p;
      g_guard_bad++;
    }
  }
  g_guards.swap(keep);
  return 0;
}

// A chain call is about to lay ITS members (and gaps) over [ws, ws + bytes): gaps recorded there by earlier calls belong to an
// older layout (another trunk sharing the temporary workspace, other sequence lengths) and are about to be overwritten
// legitimately -- inspect them now (device-wide synchronise: this is a diagnostic mode), then forget them.
int guard_begin(void* ws, int64_t bytes) {
  std::lock_guard<std::mutex> lk(g_guard_mu);
  bool any = false;
  for (char* p : g_guards) any |= (p >= (char*)ws && p < (char*)ws + bytes);
  if (!any) return 0;
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    vj_set_error("ws_guard: %s", hipGetErrorString(e));
    return (int)e;
  }
  return inspect_gaps((char*)ws, (char*)ws + bytes);
}

int poison_gap(char* p, hipStream_t st) {
  hipError_t e = hipMemsetAsync(p, GUARD_PATTERN, GUARD_BYTES, st);
  if (e != hipSuccess) {
    vj_set_error("ws_guard: hipMemsetAsync failed: %s", hipGetErrorString(e));
    return (int)e;
  }
  std::lock_guard<std::mutex> lk(g_guard_mu);
  g_guards.push_back(p);
  return 0;
}
}  // namespace

// -> gaps inspected since the last call in *n_checked, those that no longer held the pattern in *n_bad; synchronises the device,
// inspects every gap still recorded and resets the counters
extern "C" int vj_ws_guard_check(int64_t* n_checked, int64_t* n_bad) {
  VJ_CHECK_ARG(n_checked != nullptr && n_bad != nullptr, "vj_ws_guard_check: null output");
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) {
    vj_set_error("vj_ws_guard_check: %s", hipGetErrorString(e));
    return (int)e;
  }
  std::lock_guard<std::mutex> lk(g_guard_mu);
  if (int rc = inspect_gaps(nullptr, (char*)UINTPTR_MAX)) return rc;
  *n_checked = g_guard_checked;
  *n_bad = g_guard_bad;
  if (g_guard_bad) vj_set_error("vj_ws_guard_check: %ld damaged gaps, the first at device address %p", (long)g_guard_bad, (void*)g_guard_first_bad);
  g_guard_checked = g_guard_bad = 0;
  g_guard_first_bad = nullptr;
  return 0;
}

// ---------------------------------------------------------------------------------------------------- event pool
// Ordering events (no timing) reused round-robin: hipStreamWaitEvent captures the record that precedes it at call time,
// so an event may be re-recorded as soon as its wait has been enqueued, which always happens inside the same chain call.
namespace {
constexpr int POOL = 1024;
struct EventPool {   // one per device: an event belongs to the device that was current when it was created
  hipEvent_t ev[POOL];
  std::once_flag once;
  std::atomic<unsigned> next{0};
  bool ok = false;
};
EventPool g_pools[VJ_MAX_DEVICES];
thread_local bool g_pool_ok = false;   // of the pool this thread used last (read right after next_event())

hipEvent_t next_event() {
  EventPool& P = g_pools[vj_device_slot()];
  std::call_once(P.once, [&P] {
    P.ok = true;
    for (int i = 0; i < POOL; i++)
      if (hipEventCreateWithFlags(&P.ev[i], hipEventDisableTiming) != hipSuccess) P.ok = false;
  });
  g_pool_ok = P.ok;
  return P.ev[P.next.fetch_add(1) % POOL];
}

// `to` waits for everything enqueued so far on `from`
int stream_after(hipStream_t to, hipStream_t from, const char* what) {
  hipEvent_t e = next_event();
  if (!g_pool_ok) {
    vj_set_error("%s: could not create ordering events", what);
    return -2;
  }
  HIPCH(hipEventRecord(e, from), what);
  HIPCH(hipStreamWaitEvent(to, e, 0), what);
  return 0;
}

// ---------------------------------------------------------------------------------------------------- profiler
// bench.py's roofline object needs per-launch durations of the dominant kernels measured with HIP events on the
// launch stream.  Off by default (no event is created or recorded); vj_prof_enable(1) starts collecting.
struct ProfRec {
  hipEvent_t s, e;
  int family;   // 0 GEMM, 1 attention forward, 2 attention backward
  double flop;
  int64_t m, n, k;
  int tag;      // epilogue for GEMMs, head_dim for attention
};
std::mutex g_prof_mu;
std::vector<ProfRec> g_prof;
std::atomic<int> g_prof_on{0};

struct ProfScope {
  bool on;
  hipEvent_t s, e;
  hipStream_t st;
  int family, tag;
  double flop;
  int64_t m, n, k;
  ProfScope(hipStream_t stream, int fam, double fl, int64_t M, int64_t N, int64_t K, int tg)
      : on(g_prof_on.load() != 0), st(stream), family(fam), tag(tg), flop(fl), m(M), n(N), k(K) {
    if (!on) return;
    if (hipEventCreate(&s) != hipSuccess || hipEventCreate(&e) != hipSuccess) {
      on = false;
      return;
    }
    (void)hipEventRecord(s, st);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(e, st);
    std::lock_guard<std::mutex> lk(g_prof_mu);
    g_prof.push_back({s, e, family, flop, m, n, k, tag});
  }
};
}  // namespace

extern "C" int vj_prof_enable(int on) {
  g_prof_on.store(on ? 1 : 0);
  return 0;
}

// Sums per family: ms[3], flop[3], launches[3]; optional CSV of every launch (family,tag,m,n,k,us) at csv_path.
// Synchronises the recorded events (call after the work has been enqueued); clears the records.
extern "C" int vj_prof_collect(double* ms, double* flop, int64_t* launches, const char* csv_path) {
  std::lock_guard<std::mutex> lk(g_prof_mu);
  for (int i = 0; i < 3; i++) {
    ms[i] = 0.0;
    flop[i] = 0.0;
    launches[i] = 0;
  }
  FILE* f = csv_path && csv_path[0] ? fopen(csv_path, "w") : nullptr;
  if (f) fprintf(f, "family,tag,m,n,k,us,flop\n");
  for (auto& r : g_prof) {
    float t = 0.f;
    HIPCH(hipEventSynchronize(r.e), "vj_prof_collect");
    HIPCH(hipEventElapsedTime(&t, r.s, r.e), "vj_prof_collect");
    ms[r.family] += t;
    flop[r.family] += r.flop;
    launches[r.family] += 1;
    if (f) fprintf(f, "%d,%d,%ld,%ld,%ld,%.3f,%.6e\n", r.family, r.tag, (long)r.m, (long)r.n, (long)r.k, 1e3 * t, r.flop);
    (void)hipEventDestroy(r.s);
    (void)hipEventDestroy(r.e);
  }
  if (f) fclose(f);
  g_prof.clear();
  return 0;
}

// ---------------------------------------------------------------------------------------------------- launch helpers
// GEMM kernel-selection flags per role: run-time options gemm_fwd_flags / gemm_dgrad_flags (options.hpp), e.g. 256 = gemm4w.hip
static int gemm(const void* A, int64_t lda, const void* B, int64_t ldb, void* C, int64_t ldc, int64_t M, int64_t N,
                int64_t K, const float* bias, const void* res, int64_t ldr, const void* aux_in, void* aux_out,
                int64_t ldaux, int epi, hipStream_t st, int flags = 0) {
  ProfScope ps(st, 0, 2.0 * M * N * K, M, N, K, epi);
  return vj_gemm_bf16_nt(A, lda, B, ldb, C, ldc, M, N, K, bias, res, ldr, aux_in, aux_out, ldaux, epi, 1.0f, 0.0f, flags, st);
}

struct FwdLayout {
  int64_t x, y1, qkv, o, x1, y2, u, g, mean1, rstd1, mean2, rstd2, lse, rs, total;
  int64_t gap[16];   // option ws_guard: offsets of the 256-byte gaps behind the members
  int n_gap;
};
static FwdLayout fwd_layout(int64_t M, int64_t D, int64_t Dh, int64_t H) {
  FwdLayout L;
  L.n_gap = 0;
  const int64_t gg = guard_gap();
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    int64_t o = off;
    off += al256(bytes);
    if (gg) {
      L.gap[L.n_gap++] = off;
      off += gg;
    }
    return o;
  };
  L.x = take(M * D * 2);
  L.y1 = take(M * D * 2);
  L.qkv = take(M * 3 * D * 2);
  L.o = take(M * D * 2);
  L.x1 = take(M * D * 2);
  L.y2 = take(M * D * 2);
  L.u = take(M * Dh * 2);
  L.g = take(M * Dh * 2);
  L.mean1 = take(M * 4);
  L.rstd1 = take(M * 4);
  L.mean2 = take(M * 4);
  L.rstd2 = take(M * 4);
  L.lse = take(H * M * 4);
  L.rs = take(M * 8);   // {rstd, -mean * rstd} per row: the LayerNorm folded into the consuming GEMM (vj_blocks_fwd_lnfold)
  L.total = off;
  return L;
}

extern "C" int64_t vj_blocks_fwd_ws_bytes(int64_t M, int64_t D, int64_t Dh, int64_t heads, int64_t n_blocks, int save) {
  const FwdLayout L = fwd_layout(M, D, Dh, heads);
  return save ? L.total * n_blocks : L.total + al256(M * D * 2) + guard_gap();
}

static int check_blocks(const vj_block_t* blocks, int64_t n_blocks, int64_t D, const char* who) {
  VJ_CHECK_ARG(blocks != nullptr && n_blocks > 0, "%s: no blocks", who);
  const int64_t Dh = blocks[0].fc1.n_out;
  for (int64_t i = 0; i < n_blocks; i++) {
    const vj_block_t& b = blocks[i];
    VJ_CHECK_ARG(b.qkv.n_out == 3 * D && b.qkv.k_in == D && b.proj.n_out == D && b.proj.k_in == D && b.fc1.n_out == Dh &&
                     b.fc1.k_in == D && b.fc2.n_out == D && b.fc2.k_in == Dh,
                 "%s: block %ld has inconsistent Linear shapes for D=%ld", who, (long)i, (long)D);
  }
  return 0;
}

static int check_segs(const vj_seg_t* segs, int64_t n_segs, int64_t M, const char* who) {
  VJ_CHECK_ARG(segs != nullptr && n_segs > 0, "%s: no segments", who);
  int64_t r = 0;
  for (int64_t i = 0; i < n_segs; i++) {
    VJ_CHECK_ARG(segs[i].row0 == r && segs[i].B >= 0 && segs[i].S >= 0, "%s: segments must tile the rows in order", who);
    r += segs[i].B * segs[i].S;
  }
  VJ_CHECK_ARG(r == M, "%s: segments cover %ld rows, M=%ld", who, (long)r, (long)M);
  return 0;
}

// ---------------------------------------------------------------------------------------------------- forward
int vj_ln_rowstats(const void* x_bf16, float* rowstats, int64_t rows, int64_t D, float eps, hipStream_t stream);   // norm_loss.hip
int vj_gemm_bf16_nt_lnfold(const void* X, int64_t ldx, const void* Wf, int64_t ldw, void* C, int64_t ldc, int64_t M, int64_t N,
                           int64_t K, const float* bias_f, const float* rowstats, const float* colsum_w, int epilogue, float alpha,
                           int flags, hipStream_t stream);   // gemm.hip

static int blocks_fwd_impl(const vj_block_t* blocks, const vj_lnfold_t* folds, int64_t n_blocks, const void* x_in, void* x_out, int64_t M,
                           int64_t D, int64_t heads, const vj_seg_t* segs, int64_t n_segs, float ln_eps, int save,
                           int gemm_flags, void* ws, int64_t ws_bytes, hipStream_t stream) {
  // gemm_flags: low 16 bits = kernel selection of vj_gemm_bf16_nt (0: option gemm_fwd_flags); bits 16-23 = first block the
  // selection applies to (earlier blocks take the automatic choice) -- the EMA target encoder's late blocks run after the
  // context branch has left the GPU, where the two-workgroups-per-CU kernel (0x100) is the faster one
  const int sel_flags = (gemm_flags & 0xffff) ? (gemm_flags & 0xffff) : vj_opt(VJ_OPT_GEMM_FWD_FLAGS);
  const int64_t sel_from = (gemm_flags >> 16) & 0xff;
  CH(check_blocks(blocks, n_blocks, D, "vj_blocks_fwd"));
  CH(check_segs(segs, n_segs, M, "vj_blocks_fwd"));
  VJ_CHECK_ARG(heads > 0 && D % heads == 0, "vj_blocks_fwd: D=%ld not divisible by heads=%ld", (long)D, (long)heads);
  if (M == 0) return 0;
  const int64_t Dh = blocks[0].fc1.n_out, hd = D / heads;
  const float scale = (float)pow((double)hd, -0.5);   // head_dim ** -0.5 exactly as Attention.scale (modules.py:53) is computed on the host
  VJ_CHECK_ARG(ws != nullptr && ws_bytes >= vj_blocks_fwd_ws_bytes(M, D, Dh, heads, n_blocks, save),
               "vj_blocks_fwd: workspace too small (%ld < %ld)", (long)ws_bytes,
               (long)vj_blocks_fwd_ws_bytes(M, D, Dh, heads, n_blocks, save));
  VJ_CHECK_ARG(((uintptr_t)ws & 255) == 0, "vj_blocks_fwd: workspace must be 256-byte aligned");
  const FwdLayout L = fwd_layout(M, D, Dh, heads);
  if (L.n_gap) {   // option ws_guard
    CH(guard_begin(ws, vj_blocks_fwd_ws_bytes(M, D, Dh, heads, n_blocks, save)));
    for (int64_t li = 0; li < (save ? n_blocks : 1); li++)
      for (int k = 0; k < L.n_gap; k++) CH(poison_gap((char*)ws + li * L.total + L.gap[k], stream));
    if (!save) CH(poison_gap((char*)ws + L.total + al256(M * D * 2), stream));
  }
  const bool merge_segs = n_segs > 1 && n_segs <= 4;   // all masks of the batch in ONE attention launch (profiles/r04_abab_attn_merge.md)
  const bool qpre = vj_opt(VJ_OPT_ATTN_SOFTMAX) == 2 && (3 * D) % 12 == 0;
  const float ascale = qpre ? -scale : scale;   // negative: "q is pre-scaled" (vj_attn_fwd_segs)
  char* base = (char*)ws;
  char* pingpong[2] = {base + L.x, base + L.total};   // save = 0: block outputs alternate between these two
  const char* x = (const char*)x_in;
  for (int64_t li = 0; li < n_blocks; li++) {
    const vj_block_t& b = blocks[li];
    const int fwd_flags = li >= sel_from ? sel_flags : vj_opt(VJ_OPT_GEMM_FWD_FLAGS);
    char* w = save ? base + li * L.total : base;
    char* x2;
    if (li == n_blocks - 1) x2 = (char*)x_out;
    else if (save) x2 = base + (li + 1) * L.total + L.x;
    else x2 = pingpong[li & 1];
    float* mean1 = save ? (float*)(w + L.mean1) : nullptr;
    float* rstd1 = save ? (float*)(w + L.rstd1) : nullptr;
    float* mean2 = save ? (float*)(w + L.mean2) : nullptr;
    float* rstd2 = save ? (float*)(w + L.rstd2) : nullptr;
    const vj_lnfold_t* fo = folds ? folds + li : nullptr;
    if (fo) {   // LayerNorm folded into the qkv projection: a statistics pass over x (read only), the GEMM reads x itself
      float* rs = (float*)(w + L.rs);
      CH(vj_ln_rowstats(x, rs, M, D, ln_eps, stream));
      ProfScope ps(stream, 0, 2.0 * M * 3 * D * D, M, 3 * D, D, 0);
      CH(vj_gemm_bf16_nt_lnfold(x, D, fo->w_qkv, D, w + L.qkv, 3 * D, M, 3 * D, D, fo->b_qkv, rs, fo->c_qkv, qpre ? 4 : 0,
                                qpre ? scale * 1.4426950408889634f : 1.0f, fwd_flags, stream));
    } else {
    CH(vj_layernorm_fwd(x, b.norm1.g, b.norm1.b, w + L.y1, mean1, rstd1, M, D, ln_eps, stream));
    if (qpre) {   // option attn_softmax = 2: the q third carries scale * log2(e), applied before the bf16 rounding (epilogue 4)
      ProfScope ps(stream, 0, 2.0 * M * 3 * D * D, M, 3 * D, D, 0);
      CH(vj_gemm_bf16_nt(w + L.y1, D, b.qkv.w, D, w + L.qkv, 3 * D, M, 3 * D, D, b.qkv.b, nullptr, 0, nullptr, nullptr, 0, 4,
                         scale * 1.4426950408889634f, 0.0f, fwd_flags, stream));
    } else {
      CH(gemm(w + L.y1, D, b.qkv.w, D, w + L.qkv, 3 * D, M, 3 * D, D, b.qkv.b, nullptr, 0, nullptr, nullptr, 0, 0, stream, fwd_flags));
    }
    }
    if (merge_segs) {   // all segments (masks) in ONE launch: the short one's workgroups fill the long one's tail
      double fl = 0;
      int64_t smax = 0;
      for (int64_t s = 0; s < n_segs; s++) {
        fl += 4.0 * segs[s].B * heads * segs[s].S * segs[s].S * hd;
        if (segs[s].S > smax) smax = segs[s].S;
      }
      ProfScope ps(stream, 1, fl, segs[0].B, smax, heads, (int)hd);
      CH(vj_attn_fwd_segs(w + L.qkv, w + L.o, save ? (float*)(w + L.lse) : nullptr, segs, n_segs, heads, hd, ascale, stream));
    } else {
      for (int64_t s = 0; s < n_segs; s++) {
        const vj_seg_t& sg = segs[s];
        if (sg.B * sg.S == 0) continue;
        float* lse = save ? (float*)(w + L.lse) + heads * sg.row0 : nullptr;
        ProfScope ps(stream, 1, 4.0 * sg.B * heads * sg.S * sg.S * hd, sg.B, sg.S, heads, (int)hd);
        CH(vj_attn_fwd(w + L.qkv + sg.row0 * 3 * D * 2, w + L.o + sg.row0 * D * 2, lse, sg.B, sg.S, heads, hd, ascale,
                       stream));
      }
    }
    CH(gemm(w + L.o, D, b.proj.w, D, w + L.x1, D, M, D, D, b.proj.b, x, D, nullptr, nullptr, 0, 0, stream, fwd_flags));
    if (fo) {   // LayerNorm folded into fc1 (GELU epilogue; no saved derivative: these blocks never run backward)
      float* rs = (float*)(w + L.rs);
      CH(vj_ln_rowstats(w + L.x1, rs, M, D, ln_eps, stream));
      ProfScope ps(stream, 0, 2.0 * M * Dh * D, M, Dh, D, 1);
      CH(vj_gemm_bf16_nt_lnfold(w + L.x1, D, fo->w_fc1, D, w + L.g, Dh, M, Dh, D, fo->b_fc1, rs, fo->c_fc1, 1, 1.0f, fwd_flags, stream));
    } else {
    CH(vj_layernorm_fwd(w + L.x1, b.norm2.g, b.norm2.b, w + L.y2, mean2, rstd2, M, D, ln_eps, stream));
    CH(gemm(w + L.y2, D, b.fc1.w, D, w + L.g, Dh, M, Dh, D, b.fc1.b, nullptr, 0, nullptr, save ? w + L.u : nullptr, Dh,
            1, stream, fwd_flags));
    }
    CH(gemm(w + L.g, Dh, b.fc2.w, Dh, x2, D, M, D, Dh, b.fc2.b, w + L.x1, D, nullptr, nullptr, 0, 0, stream, fwd_flags));
    x = x2;
  }
  return 0;
}

extern "C" int vj_blocks_fwd(const vj_block_t* blocks, int64_t n_blocks, const void* x_in, void* x_out, int64_t M,
                             int64_t D, int64_t heads, const vj_seg_t* segs, int64_t n_segs, float ln_eps, int save,
                             int gemm_flags, void* ws, int64_t ws_bytes, hipStream_t stream) {
  return blocks_fwd_impl(blocks, nullptr, n_blocks, x_in, x_out, M, D, heads, segs, n_segs, ln_eps, save, gemm_flags, ws, ws_bytes, stream);
}

// The same trunk with both LayerNorms of every block folded into the Linear that consumes them (vj_gemm_bf16_nt_lnfold): for
// blocks that never run backward (save must be 0) -- the EMA target encoder, frozen-encoder inference.  folds[i] holds block
// i's folded qkv / fc1 weights (vj_ln_fold_weights); the blocks' own norm / qkv / fc1 weights are not read.  Workspace as
// vj_blocks_fwd_ws_bytes(..., save = 0).
extern "C" int vj_blocks_fwd_lnfold(const vj_block_t* blocks, const vj_lnfold_t* folds, int64_t n_blocks, const void* x_in, void* x_out,
                                    int64_t M, int64_t D, int64_t heads, const vj_seg_t* segs, int64_t n_segs, float ln_eps,
                                    int gemm_flags, void* ws, int64_t ws_bytes, hipStream_t stream) {
  VJ_CHECK_ARG(folds != nullptr, "vj_blocks_fwd_lnfold: no folded weights");
  for (int64_t i = 0; i < n_blocks; i++)
    VJ_CHECK_ARG(folds[i].w_qkv && folds[i].c_qkv && folds[i].b_qkv && folds[i].w_fc1 && folds[i].c_fc1 && folds[i].b_fc1,
                 "vj_blocks_fwd_lnfold: block %ld lacks folded weights", (long)i);
  return blocks_fwd_impl(blocks, folds, n_blocks, x_in, x_out, M, D, heads, segs, n_segs, ln_eps, 0, gemm_flags, ws, ws_bytes, stream);
}

// ---------------------------------------------------------------------------------------------------- backward
#define WGRAD_WS_BYTES ((int64_t)96 << 20)
#define GROUP_WS_BYTES ((int64_t)192 << 20)   // grouped weight gradients: 4 * sum N1*N2 (ViT-H: 79 MB) x split factor

struct BwdLayout {
  int64_t du[2], dx1[2], dqkv[2], dx[3], dy2, dob, dy1, delta, ln_ws, ln_ws2, colp_fc1, colp_q, colp_kv, dyT[2], xT[2], tcs_ws[2],
      wg_ws[2], total;
  int64_t ln_ws_bytes, tcs_ws_bytes, delta_bytes, colp_fc1_rows, colp_attn_rows;
  int64_t gap[32];   // option ws_guard
  int n_gap;
};
int vj_layernorm_bwd_partials(const void* dy_bf16, const void* x_bf16, const float* gamma, const float* mean, const float* rstd,
                              const void* dres_bf16, void* dx_bf16, bool cs, int64_t rows, int64_t D, void* ws,
                              int64_t ws_bytes, int64_t* nb_out, hipStream_t stream);   // norm_loss.hip
static BwdLayout bwd_layout(int64_t M, int64_t D, int64_t Dh, int64_t H) {
  BwdLayout L;
  L.n_gap = 0;
  const int64_t gg = guard_gap();
  int64_t off = 0;
  auto take = [&](int64_t bytes) {
    int64_t o = off;
    off += al256(bytes);
    if (gg) {
      L.gap[L.n_gap++] = off;
      off += gg;
    }
    return o;
  };
  const int64_t Mp = pad64i(M), nmax = 3 * D > Dh ? 3 * D : Dh;
  for (int p = 0; p < 2; p++) {
    L.du[p] = take(M * Dh * 2);
    L.dx1[p] = take(M * D * 2);
    L.dqkv[p] = take(M * 3 * D * 2);
  }
  for (int p = 0; p < 3; p++) L.dx[p] = take(M * D * 2);
  L.dy2 = take(M * D * 2);
  L.dob = take(M * D * 2);
  L.dy1 = take(M * D * 2);
  L.delta_bytes = H * M * 4;
  L.delta = take(L.delta_bytes);
  L.ln_ws_bytes = vj_layernorm_bwd_ws_bytes(D);
  L.ln_ws = take(L.ln_ws_bytes);
  // option bias_fuse: the block's column partials stay alive until ONE reduction at the end of the block's backward --
  // a second LayerNorm partial buffer (norm2's), the fc2-dgrad epilogue's sums of du (fc1 bias) and the attention backward's
  // sums of dqkv (qkv bias; rows bounded by M/8 + 16: enough for sequences of >= 10 tokens, shorter ones take the unfused route)
  L.ln_ws2 = take(L.ln_ws_bytes);
  L.colp_fc1_rows = vj_gemm_colsum_rows(M);
  L.colp_fc1 = take(L.colp_fc1_rows * Dh * 4);
  L.colp_attn_rows = M / 8 + 16;
  L.colp_q = take(L.colp_attn_rows * D * 4);
  L.colp_kv = take(L.colp_attn_rows * 2 * D * 4);
  L.tcs_ws_bytes = vj_transpose_colsum_ws_bytes(M, nmax);
  if (vj_colsum_ws_bytes(nmax) > L.tcs_ws_bytes) L.tcs_ws_bytes = vj_colsum_ws_bytes(nmax);
  for (int w = 0; w < 1; w++) {   // scratch of the weight-gradient stream
    L.dyT[w] = take(nmax * Mp * 2);
    L.xT[w] = take(Dh * Mp * 2);
    L.tcs_ws[w] = take(L.tcs_ws_bytes);
    L.wg_ws[w] = take(w == 0 ? GROUP_WS_BYTES : WGRAD_WS_BYTES);   // lane 0 also serves the grouped launch
  }
  L.total = off;
  return L;
}

extern "C" int64_t vj_blocks_bwd_ws_bytes(int64_t M, int64_t D, int64_t Dh, int64_t heads) {
  return bwd_layout(M, D, Dh, heads).total;
}

struct SideCtx {
  hipStream_t main, side;   // side == main: serial mode
  char* tmp;
  const BwdLayout* L;
  int64_t M;
  float alpha, beta;
  int tn;
};

// dW (fp32, += beta*old) = alpha * dy^T x_in ; db = alpha * colsum(dy) -- on the side stream, after `main` produced dy
static int wgrad(const SideCtx& c, const void* dy, const void* x_in, const vj_linear_t& lw_in, bool bias_done = false) {
  constexpr int lane = 0;   // one weight-gradient stream, one set of scratch buffers
  vj_linear_t lw = lw_in;
  if (bias_done) lw.gb = nullptr;   // the bias gradient (column sum of dy) came out of the LayerNorm backward that produced dy
  const int64_t M = c.M, Mp = pad64i(M), N = lw.n_out, K = lw.k_in;
  hipStream_t st = c.side;
  if (st != c.main) CH(stream_after(st, c.main, "vj_blocks_bwd(fork)"));
  if (c.tn && N % 8 == 0 && K % 8 == 0) {
    if (lw.gb) CH(vj_colsum_bf16(dy, M, N, N, M > 0 ? M : 1, 0, M > 0 ? M : 1, lw.gb, c.alpha, c.beta, c.tmp + c.L->tcs_ws[lane],
                                 c.L->tcs_ws_bytes, st));
    ProfScope ps(st, 0, 2.0 * M * N * K, N, K, M, 3);
    return vj_gemm_bf16_tn_splitk(dy, N, x_in, K, lw.gw, K, M, N, K, c.alpha, c.beta, c.tmp + c.L->wg_ws[lane], WGRAD_WS_BYTES, st);
  }
  char* dyT = c.tmp + c.L->dyT[lane];
  char* xT = c.tmp + c.L->xT[lane];
  if (lw.gb) CH(vj_transpose_colsum_bf16(dy, dyT, M, N, N, Mp, lw.gb, c.alpha, c.beta, c.tmp + c.L->tcs_ws[lane],
                                         c.L->tcs_ws_bytes, st));
  else CH(vj_transpose_bf16(dy, dyT, M, N, N, Mp, st));
  CH(vj_transpose_bf16(x_in, xT, M, K, K, Mp, st));
  ProfScope ps(st, 0, 2.0 * N * K * Mp, N, K, Mp, 3);
  return vj_gemm_bf16_nt_splitk(dyT, Mp, xT, Mp, lw.gw, K, N, K, Mp, c.alpha, c.beta, 0, c.tmp + c.L->wg_ws[lane],
                                WGRAD_WS_BYTES, st);
}

// The block's four weight gradients as ONE grouped launch (option wgrad_group): the bias column sums that no LayerNorm
// backward produced go first, then vj_gemm_bf16_tn_grouped; everything on the side stream, after `main` produced the
// last dY of the block.
struct WgradItem {
  const void* dy;
  const void* x;
  const vj_linear_t* lw;
  bool bias_done;
};
static int wgrad_group(const SideCtx& c, const WgradItem* it, int n) {
  hipStream_t st = c.side;
  if (st != c.main) CH(stream_after(st, c.main, "vj_blocks_bwd(fork)"));
  vj_tn_problem_t pr[4];
  double fl = 0;
  int64_t out_elems = 0;
  for (int i = 0; i < n; i++) {
    const vj_linear_t& lw = *it[i].lw;
    const int64_t N = lw.n_out, K = lw.k_in, M = c.M;
    if (lw.gb && !it[i].bias_done)
      CH(vj_colsum_bf16(it[i].dy, M, N, N, M > 0 ? M : 1, 0, M > 0 ? M : 1, lw.gb, c.alpha, c.beta, c.tmp + c.L->tcs_ws[0],
                        c.L->tcs_ws_bytes, st));
    pr[i] = vj_tn_problem_t{it[i].dy, N, it[i].x, K, lw.gw, K, N, K};
    fl += 2.0 * M * N * K;
    out_elems += N * K;
  }
  ProfScope ps(st, 0, fl, out_elems / pr[0].N2, pr[0].N2, c.M, 4);
  return vj_gemm_bf16_tn_grouped(pr, n, c.M, c.alpha, c.beta, c.tmp + c.L->wg_ws[0], GROUP_WS_BYTES, st);
}

extern "C" int vj_blocks_bwd(const vj_block_t* blocks, int64_t n_blocks, const void* x_in, const void* dout, void* dx_out,
                             int64_t M, int64_t D, int64_t heads, const vj_seg_t* segs, int64_t n_segs, float alpha,
                             float beta_acc, const void* save_ws, int64_t save_ws_bytes, void* tmp_ws,
                             int64_t tmp_ws_bytes, int flags, hipStream_t stream, hipStream_t side,
                             vj_layer_cb_t on_layer_done, void* user) {
  CH(check_blocks(blocks, n_blocks, D, "vj_blocks_bwd"));
  CH(check_segs(segs, n_segs, M, "vj_blocks_bwd"));
  VJ_CHECK_ARG(heads > 0 && D % heads == 0, "vj_blocks_bwd: D=%ld not divisible by heads=%ld", (long)D, (long)heads);
  if (M == 0) return 0;
  const int g_dgrad_flags = vj_opt(VJ_OPT_GEMM_DGRAD_FLAGS);
  const int64_t Dh = blocks[0].fc1.n_out, hd = D / heads;
  const float scale = (float)pow((double)hd, -0.5);   // head_dim ** -0.5 exactly as Attention.scale (modules.py:53) is computed on the host
  const FwdLayout F = fwd_layout(M, D, Dh, heads);
  const BwdLayout L = bwd_layout(M, D, Dh, heads);
  VJ_CHECK_ARG(save_ws != nullptr && save_ws_bytes >= F.total * n_blocks, "vj_blocks_bwd: saved-activation workspace too small");
  VJ_CHECK_ARG(tmp_ws != nullptr && tmp_ws_bytes >= L.total, "vj_blocks_bwd: temporary workspace too small (%ld < %ld)",
               (long)tmp_ws_bytes, (long)L.total);
  VJ_CHECK_ARG((((uintptr_t)save_ws | (uintptr_t)tmp_ws) & 255) == 0, "vj_blocks_bwd: workspaces must be 256-byte aligned");
  for (int64_t i = 0; i < n_blocks; i++) {
    const vj_block_t& b = blocks[i];
    VJ_CHECK_ARG(b.qkv.wT && b.proj.wT && b.fc1.wT && b.fc2.wT && b.qkv.gw && b.proj.gw && b.fc1.gw && b.fc2.gw &&
                     b.norm1.gg && b.norm1.gb && b.norm2.gg && b.norm2.gb,
                 "vj_blocks_bwd: block %ld lacks transposed weights / gradient views", (long)i);
  }
  const char* sv = (const char*)save_ws;
  char* tmp = (char*)tmp_ws;
  if (L.n_gap) CH(guard_begin(tmp, L.total));   // option ws_guard
  for (int k = 0; k < L.n_gap; k++) CH(poison_gap(tmp + L.gap[k], stream));
  SideCtx sc{stream, side ? side : stream, tmp, &L, M, alpha, beta_acc, ((flags & 1) || vj_opt(VJ_OPT_WGRAD_TN)) ? 1 : 0};
  constexpr int MAX_BLOCKS = 256;
  VJ_CHECK_ARG(n_blocks <= MAX_BLOCKS, "vj_blocks_bwd: more than %d blocks", MAX_BLOCKS);
  hipEvent_t side_done[MAX_BLOCKS];
  const char* dx2 = (const char*)dout;
  for (int64_t li = n_blocks - 1; li >= 0; li--) {
    const vj_block_t& b = blocks[li];
    const char* w = sv + li * F.total;
    const char* x = li == 0 ? (const char*)x_in : w + F.x;
    const int p = (int)(li & 1);
    char* du = tmp + L.du[p];
    char* dx1 = tmp + L.dx1[p];
    char* dqkv = tmp + L.dqkv[p];
    char* dx = li == 0 ? (char*)dx_out : tmp + L.dx[li % 3];
    // the buffers this block is about to overwrite were last read by the weight gradients of block li+2
    if (sc.side != sc.main && li + 2 < n_blocks) HIPCH(hipStreamWaitEvent(stream, side_done[li + 2], 0), "vj_blocks_bwd");
    // fc2: dgrad fused with GELU' ; wgrad reads (dx2, g)
    // dx2 of every block but the last is the dx of block li+1's norm1 backward, which also produced its column sums
    const bool fuse_cs = sc.tn != 0;   // (the NT route folds the bias gradient into its dY transpose instead)
    const bool grouped = sc.tn != 0 && vj_opt(VJ_OPT_WGRAD_GROUP) != 0 && D % 8 == 0 && Dh % 8 == 0;
    // (flags bit 1: the caller's final-norm backward already wrote the LAST block's fc2 bias gradient, the column sums of dout)
    const bool fc2_done = fuse_cs && (li + 1 < n_blocks || (flags & 2) != 0);
    const WgradItem items[4] = {{dx2, w + F.g, &b.fc2, fc2_done},
                                {du, w + F.y2, &b.fc1, false},
                                {dx1, w + F.o, &b.proj, fuse_cs},
                                {dqkv, w + F.y1, &b.qkv, false}};
    if (!grouped) CH(wgrad(sc, dx2, w + F.g, b.fc2, fc2_done));
    // option bias_fuse (with the transpose-free route): every column partial the block produces -- both LayerNorm backwards',
    // the fc2-dgrad epilogue's sums of du (= fc1's bias gradient) and the attention backward's sums of dqkv (= qkv's) -- is
    // reduced by ONE vj_reduce_segments launch at the end of the block instead of two reductions + two column-sum passes over
    // du / dqkv + their two reductions (6 launches, 143 MB re-read per ViT-L context block)
    const bool bfuse = fuse_cs && vj_opt(VJ_OPT_BIAS_FUSE) != 0;
    vj_reduce_seg_t rsegs[12];
    int n_rsegs = 0;
    int fc1_fused = 0;
    if (bfuse && b.fc1.gb != nullptr) {
      ProfScope ps(stream, 0, 2.0 * M * Dh * D, M, Dh, D, 2);
      CH(vj_gemm_bf16_nt_dgelu_colsum(dx2, D, b.fc2.wT, b.fc2.ldwT, du, Dh, M, Dh, D, w + F.u, Dh, (float*)(tmp + L.colp_fc1),
                                      L.colp_fc1_rows, g_dgrad_flags, &fc1_fused, stream));
      if (fc1_fused) rsegs[n_rsegs++] = vj_reduce_seg_t{(const float*)(tmp + L.colp_fc1), b.fc1.gb, L.colp_fc1_rows, Dh, Dh};
    } else {
      CH(gemm(dx2, D, b.fc2.wT, b.fc2.ldwT, du, Dh, M, Dh, D, nullptr, nullptr, 0, w + F.u, nullptr, Dh, 2, stream, g_dgrad_flags));
    }
    // fc1
    if (!grouped) CH(wgrad(sc, du, w + F.y2, b.fc1, fc1_fused != 0));
    CH(gemm(du, Dh, b.fc1.wT, b.fc1.ldwT, tmp + L.dy2, D, M, D, Dh, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, stream, g_dgrad_flags));
    // dx1 is the dY of proj: its bias gradient = column sums of dx1, produced by this pass
    if (bfuse) {
      int64_t nb2 = 0;
      const bool cs2 = b.proj.gb != nullptr;
      CH(vj_layernorm_bwd_partials(tmp + L.dy2, w + F.x1, b.norm2.g, (const float*)(w + F.mean2), (const float*)(w + F.rstd2), dx2,
                                   dx1, cs2, M, D, tmp + L.ln_ws2, L.ln_ws_bytes, &nb2, stream));
      const int64_t st2 = (cs2 ? 3 : 2) * D;
      const float* p2 = (const float*)(tmp + L.ln_ws2);
      rsegs[n_rsegs++] = vj_reduce_seg_t{p2, b.norm2.gg, nb2, D, st2};
      rsegs[n_rsegs++] = vj_reduce_seg_t{p2 + D, b.norm2.gb, nb2, D, st2};
      if (cs2) rsegs[n_rsegs++] = vj_reduce_seg_t{p2 + 2 * D, b.proj.gb, nb2, D, st2};
    } else {
      CH(vj_layernorm_bwd_colsum(tmp + L.dy2, w + F.x1, b.norm2.g, (const float*)(w + F.mean2), (const float*)(w + F.rstd2), dx2,
                                 dx1, b.norm2.gg, b.norm2.gb, fuse_cs ? b.proj.gb : nullptr, alpha, beta_acc, M, D,
                                 tmp + L.ln_ws, L.ln_ws_bytes, stream));
    }
    // proj
    if (!grouped) CH(wgrad(sc, dx1, w + F.o, b.proj, fuse_cs));
    CH(gemm(dx1, D, b.proj.wT, b.proj.ldwT, tmp + L.dob, D, M, D, D, nullptr, nullptr, 0, nullptr, nullptr, 0, 0, stream, g_dgrad_flags));
    // qkv bias: column partials from the attention backward kernels, when every segment's partial rows fit the workspace
    int64_t rows_q = 0, rows_kv = 0;
    bool qkv_fused = bfuse && b.qkv.gb != nullptr;
    if (qkv_fused) {
      for (int64_t s = 0; s < n_segs; s++) {
        int64_t rq = 0, rkv = 0;
        if (segs[s].B * segs[s].S == 0) continue;
        CH(vj_attn_bwd_colsum_rows(segs[s].B, segs[s].S, hd, &rq, &rkv));
        rows_q += rq;
        rows_kv += rkv;
      }
      if (rows_q > L.colp_attn_rows || rows_kv > L.colp_attn_rows) qkv_fused = false;
    }
    int64_t off_q = 0, off_kv = 0;
    const bool merge_segs = n_segs > 1 && n_segs <= 4;
    // as the FORWARD stored q: flags bit 3 set -> bit 2 says whether q is pre-scaled (the caller recorded the mode its vj_blocks_fwd
    // call used); otherwise the option is read again, which is only right if it did not change since that forward
    const bool qpre_b = (flags & 8) ? (flags & 4) != 0 : (vj_opt(VJ_OPT_ATTN_SOFTMAX) == 2 && (3 * D) % 12 == 0);
    const float ascale = qpre_b ? -scale : scale;
    if (merge_segs) {   // one dQ + one dK/dV launch for all segments (partials: segment after segment, as the loop below lays them out)
      double fl = 0;
      int64_t smax = 0;
      for (int64_t s = 0; s < n_segs; s++) {
        fl += 8.0 * segs[s].B * heads * segs[s].S * segs[s].S * hd;
        if (segs[s].S > smax) smax = segs[s].S;
      }
      ProfScope ps(stream, 2, fl, segs[0].B, smax, heads, (int)hd);
      CH(vj_attn_bwd_segs(w + F.qkv, w + F.o, tmp + L.dob, (const float*)(w + F.lse), dqkv, segs, n_segs, heads, hd, ascale,
                          tmp + L.delta, L.delta_bytes, qkv_fused ? (float*)(tmp + L.colp_q) : nullptr,
                          qkv_fused ? (float*)(tmp + L.colp_kv) : nullptr, stream));
    }
    for (int64_t s = 0; s < n_segs && !merge_segs; s++) {
      const vj_seg_t& sg = segs[s];
      if (sg.B * sg.S == 0) continue;
      ProfScope ps(stream, 2, 8.0 * sg.B * heads * sg.S * sg.S * hd, sg.B, sg.S, heads, (int)hd);
      if (qkv_fused) {
        int64_t rq = 0, rkv = 0;
        CH(vj_attn_bwd_colsum_rows(sg.B, sg.S, hd, &rq, &rkv));
        CH(vj_attn_bwd_colsum(w + F.qkv + sg.row0 * 3 * D * 2, w + F.o + sg.row0 * D * 2, tmp + L.dob + sg.row0 * D * 2,
                              (const float*)(w + F.lse) + heads * sg.row0, dqkv + sg.row0 * 3 * D * 2, sg.B, sg.S, heads, hd,
                              ascale, tmp + L.delta, L.delta_bytes, (float*)(tmp + L.colp_q) + off_q * D,
                              (float*)(tmp + L.colp_kv) + off_kv * 2 * D, stream));
        off_q += rq;
        off_kv += rkv;
      } else {
        CH(vj_attn_bwd(w + F.qkv + sg.row0 * 3 * D * 2, w + F.o + sg.row0 * D * 2, tmp + L.dob + sg.row0 * D * 2,
                       (const float*)(w + F.lse) + heads * sg.row0, dqkv + sg.row0 * 3 * D * 2, sg.B, sg.S, heads, hd, ascale,
                       tmp + L.delta, L.delta_bytes, stream));
      }
    }
    if (qkv_fused) {
      rsegs[n_rsegs++] = vj_reduce_seg_t{(const float*)(tmp + L.colp_q), b.qkv.gb, rows_q, D, D};
      rsegs[n_rsegs++] = vj_reduce_seg_t{(const float*)(tmp + L.colp_kv), b.qkv.gb + D, rows_kv, 2 * D, 2 * D};
    }
    // qkv
    if (grouped) {
      WgradItem git[4] = {items[0], items[1], items[2], items[3]};
      git[1].bias_done = fc1_fused != 0;
      git[3].bias_done = qkv_fused;
      CH(wgrad_group(sc, git, 4));
    } else {
      CH(wgrad(sc, dqkv, w + F.y1, b.qkv, qkv_fused));
    }
    CH(gemm(dqkv, 3 * D, b.qkv.wT, b.qkv.ldwT, tmp + L.dy1, D, M, D, 3 * D, nullptr, nullptr, 0, nullptr, nullptr, 0, 0,
            stream, g_dgrad_flags));
    // dx is the dY of the previous block's fc2 (its dx2): that bias gradient comes out of this pass
    if (bfuse) {
      int64_t nb1 = 0;
      float* prev_gb = li > 0 ? blocks[li - 1].fc2.gb : nullptr;
      const bool cs1 = prev_gb != nullptr;
      CH(vj_layernorm_bwd_partials(tmp + L.dy1, x, b.norm1.g, (const float*)(w + F.mean1), (const float*)(w + F.rstd1), dx1, dx,
                                   cs1, M, D, tmp + L.ln_ws, L.ln_ws_bytes, &nb1, stream));
      const int64_t st1 = (cs1 ? 3 : 2) * D;
      const float* p1 = (const float*)(tmp + L.ln_ws);
      rsegs[n_rsegs++] = vj_reduce_seg_t{p1, b.norm1.gg, nb1, D, st1};
      rsegs[n_rsegs++] = vj_reduce_seg_t{p1 + D, b.norm1.gb, nb1, D, st1};
      if (cs1) rsegs[n_rsegs++] = vj_reduce_seg_t{p1 + 2 * D, prev_gb, nb1, D, st1};
      CH(vj_reduce_segments(rsegs, n_rsegs, alpha, beta_acc, stream));   // the block's ONE reduction launch
    } else {
      CH(vj_layernorm_bwd_colsum(tmp + L.dy1, x, b.norm1.g, (const float*)(w + F.mean1), (const float*)(w + F.rstd1), dx1, dx,
                                 b.norm1.gg, b.norm1.gb, (fuse_cs && li > 0) ? blocks[li - 1].fc2.gb : nullptr, alpha, beta_acc,
                                 M, D, tmp + L.ln_ws, L.ln_ws_bytes, stream));
    }
    if (sc.side != sc.main) {
      hipEvent_t e = next_event();
      HIPCH(hipEventRecord(e, sc.side), "vj_blocks_bwd");
      side_done[li] = e;
    }
    if (on_layer_done) on_layer_done(user, (int)li);
    dx2 = dx;
  }
  return 0;
}
