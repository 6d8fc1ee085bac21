// Dense GEMM entry point with a per-shape plan cache over hipBLASLt.
//
// Why this exists: the training step is launch-bound on the host.  Through ATen every GEMM costs
// ~17-18 us of host time on this stack (descriptor + layout + preference creation and a heuristic query per call,
// measured with tools/host_gemm_cost.py) against ~4 us for an elementwise kernel, and the step issues ~640 GEMMs (1x1
// convs of the ResNet body as GEMMs, the transformer's token projections, their input- and weight-gradients).  The
// shapes repeat every step, so the plan (matmul descriptor, three layouts, chosen algorithm) is built ONCE per distinct
// (types, transposes, sizes, leading dimensions, batch, epilogue) and a call is: hash lookup + hipblasLtMatmul.
// The math is hipBLASLt's own kernels (the same ones ATen would run): this file is runtime plumbing, not a kernel.
//
// Row-major semantics (what the Python side holds):  C[M,N] = alpha * op(A) * op(B) + beta * C  (+ bias[N] per row)
//   op(A) is [M,K]: A stored [M,K] (ld = lda) or, with transA, stored [K,M];  op(B) is [K,N]: B stored [K,N] or [N,K].
// hipBLASLt is column-major: a row-major X[r,c] (ld) is the column-major matrix X^T [c,r] (ld), so the call computes
// C^T = op(B)^T op(A)^T with the operands swapped; the bias epilogue broadcasts along C^T's rows = C's columns.
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <hipblaslt/hipblaslt-ext.hpp>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <unordered_map>

#include "../../include/ocpg_hip.h"
#include "fill.h"

namespace {

struct Key {
  int dtype, out_dtype, ta, tb, bias, beta0, epi;      // epi: bit0 ReLU, bit1 per-column alpha vector, bit2 fp32 bias
  int64_t m, n, k, lda, ldb, ldc, batch, sa, sb, sc;
  bool operator==(const Key& o) const {
    return dtype == o.dtype && out_dtype == o.out_dtype && ta == o.ta && tb == o.tb && bias == o.bias && beta0 == o.beta0 && epi == o.epi && m == o.m &&
           n == o.n && k == o.k && lda == o.lda && ldb == o.ldb && ldc == o.ldc && batch == o.batch && sa == o.sa && sb == o.sb && sc == o.sc;
  }
};
struct KeyHash {
  size_t operator()(const Key& k) const {
    uint64_t h = 1469598103934665603ull;
    auto mix = [&h](uint64_t v) { h = (h ^ v) * 1099511628211ull; };
    mix(k.dtype); mix(k.out_dtype); mix(k.ta); mix(k.tb); mix(k.bias); mix(k.beta0); mix(k.epi); mix(k.m); mix(k.n); mix(k.k); mix(k.lda); mix(k.ldb);
    mix(k.ldc); mix(k.batch); mix(k.sa); mix(k.sb); mix(k.sc);
    return (size_t)h;
  }
};
constexpr int kRanked = 12;               // candidates taken from the heuristic's ranked list
constexpr int kCandidates = 160;          // ... + (OCPG_GEMM_TUNE_ALL=1) every other kernel of the library that supports the problem
struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t workspace = 0;
  int status = 0;
  // the heuristic's ranked candidates; the first call that can be repeated without changing its result times them on the real
  // operands and keeps the fastest (see tune())
  hipblasLtMatmulAlgo_t cand[kCandidates];
  size_t cand_ws[kCandidates];
  int ncand = 0, picked = 0;
  bool tuned = false;
};

constexpr size_t kWorkspaceBytes = 64u << 20;
constexpr size_t kMaxPlans = 4096;        // variable-resolution training adds a plan per distinct M per layer: bounded
constexpr int kMaxDevices = 64;

// One state PER DEVICE (handle, plans) and one workspace per (device, stream): a process that drives a second GPU, or
// issues GEMMs on two streams, never shares a workspace or hands a pointer of another device to hipBLASLt.
struct State {
  std::mutex mu;
  hipblasLtHandle_t handle = nullptr;
  hipEvent_t ev0 = nullptr, ev1 = nullptr;
  void* cmp_ref = nullptr;            // the reference candidate's whole output (grown on demand)
  size_t cmp_bytes = 0;
  void* scratch_c = nullptr;          // output stand-in for tuning a plan that accumulates into its C (beta != 0)
  size_t scratch_bytes = 0;
  unsigned* cmp_out = nullptr;        // [2]: bit patterns of max |c - ref| and max |ref|
  long long tuned_plans = 0, tuned_changed = 0, tuned_rejected = 0;
  std::unordered_map<hipStream_t, void*> workspaces;
  std::unordered_map<Key, Plan, KeyHash> plans;
  std::unordered_map<uint64_t, int> imported;      // key hash -> candidate index chosen by ANOTHER rank (ocpg_gemm_import_picks)
};
State* state() {
  static State table[kMaxDevices];
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMaxDevices) return nullptr;
  return &table[dev];
}

void destroy(Plan& p) {
  if (p.a) hipblasLtMatrixLayoutDestroy(p.a);
  if (p.b) hipblasLtMatrixLayoutDestroy(p.b);
  if (p.c) hipblasLtMatrixLayoutDestroy(p.c);
  if (p.desc) hipblasLtMatmulDescDestroy(p.desc);
}

// handle + this stream's workspace (call with s.mu held); 0 or an error code
int prepare(State& s, hipStream_t st, void** ws) {
  if (!s.handle && hipblasLtCreate(&s.handle) != HIPBLAS_STATUS_SUCCESS) return -1100;
  auto it = s.workspaces.find(st);
  if (it == s.workspaces.end()) {
    void* w = nullptr;
    if (hipMalloc(&w, kWorkspaceBytes) != hipSuccess) return -1099;
    it = s.workspaces.emplace(st, w).first;
  }
  *ws = it->second;
  return 0;
}

Plan build(State& s, const Key& key);

bool tuning_fp32() {      // experiment switch (default off): also time fp32 plans, validated to 1e-5 of max |C| per element.  Round 4 tried it as
  static const bool on = [] {        // the default with every supporting kernel as a candidate (-0.18 ms per step): one run of the fp32 case of
    const char* e = getenv("OCPG_GEMM_TUNE_FP32");      // test_fused_linear_bias_relu_dropout then came out 2e-3 off -- a candidate that had passed the validation
    return e && e[0] == '1';                            // and was wrong later.  fp32 is the parity-critical path (MSDeformAttn locations): default off again.
  }();
  return on;
}

int g_tuning_override = -1;       // ocpg_gemm_set_tuning: -1 = the environment decides
bool tune_all() {
  static const bool on = [] { const char* e = getenv("OCPG_GEMM_TUNE_ALL"); return !(e && e[0] == '0'); }();      // default on (round 4): -1.0 ms per step
  return on;
}

bool tuning() {
  static const bool on = [] {
    const char* e = getenv("OCPG_GEMM_TUNE");
    return !(e && e[0] == '0');
  }();
  return g_tuning_override < 0 ? on : g_tuning_override != 0;
}

Plan& plan_for(State& s, const Key& key) {
  auto it = s.plans.find(key);
  if (it == s.plans.end()) {
    if (s.plans.size() >= kMaxPlans) {          // simplest bounded policy: start over (plans are rebuilt on demand)
      for (auto& kv : s.plans) destroy(kv.second);
      s.plans.clear();
    }
    it = s.plans.emplace(key, build(s, key)).first;
    auto im = s.imported.find((uint64_t)KeyHash()(key));
    if (im != s.imported.end()) {       // the tuning rank's choice for this shape (the ranked candidate list is the same on every rank)
      Plan& p = it->second;
      if (im->second >= 0 && im->second < p.ncand) { p.algo = p.cand[im->second]; p.workspace = p.cand_ws[im->second]; p.picked = im->second; }
      p.tuned = true;
    }
  }
  return it->second;
}

hipDataType hip_type(int dtype) { return dtype == 0 ? HIP_R_32F : dtype == 1 ? HIP_R_16BF : HIP_R_16F; }

int set_batch(hipblasLtMatrixLayout_t l, int64_t batch, int64_t stride) {
  if (batch <= 1) return 0;
  int32_t b = (int32_t)batch;
  if (hipblasLtMatrixLayoutSetAttribute(l, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &b, sizeof(b)) != HIPBLAS_STATUS_SUCCESS) return -1;
  if (hipblasLtMatrixLayoutSetAttribute(l, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &stride, sizeof(stride)) != HIPBLAS_STATUS_SUCCESS)
    return -1;
  return 0;
}

// builds the plan for `key` (column-major call: m' = n, n' = m, A' = B, B' = A)
Plan build(State& s, const Key& key) {
  Plan p;
  const hipDataType in = hip_type(key.dtype), out = hip_type(key.out_dtype);
  if (hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) { p.status = -1101; return p; }
  const hipblasOperation_t opa = key.tb ? HIPBLAS_OP_T : HIPBLAS_OP_N, opb = key.ta ? HIPBLAS_OP_T : HIPBLAS_OP_N;
  hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSA, &opa, sizeof(opa));
  hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_TRANSB, &opb, sizeof(opb));
  if (key.bias || (key.epi & 1)) {
    const hipblasLtEpilogue_t ep = key.bias ? ((key.epi & 1) ? HIPBLASLT_EPILOGUE_RELU_BIAS : HIPBLASLT_EPILOGUE_BIAS) : HIPBLASLT_EPILOGUE_RELU;
    hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_EPILOGUE, &ep, sizeof(ep));
    if (key.bias) {
      const hipDataType bt = (key.epi & 4) ? HIP_R_32F : in;
      hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_DATA_TYPE, &bt, sizeof(bt));
    }
  }
  if (key.epi & 2) {      // alpha = device vector over the rows of the column-major D = the columns (output channels) of C
    const int32_t pm = HIPBLASLT_POINTER_MODE_ALPHA_DEVICE_VECTOR_BETA_HOST;
    if (hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_POINTER_MODE, &pm, sizeof(pm)) != HIPBLAS_STATUS_SUCCESS) { p.status = -1106; return p; }
  }
  // A' = B: stored row-major [K,N] (ldb) = column-major [N,K]; with transB stored [N,K] = column-major [K,N]
  const int64_t a_rows = key.tb ? key.k : key.n, a_cols = key.tb ? key.n : key.k;
  const int64_t b_rows = key.ta ? key.m : key.k, b_cols = key.ta ? key.k : key.m;
  if (hipblasLtMatrixLayoutCreate(&p.a, in, a_rows, a_cols, key.ldb) != HIPBLAS_STATUS_SUCCESS ||
      hipblasLtMatrixLayoutCreate(&p.b, in, b_rows, b_cols, key.lda) != HIPBLAS_STATUS_SUCCESS ||
      hipblasLtMatrixLayoutCreate(&p.c, out, key.n, key.m, key.ldc) != HIPBLAS_STATUS_SUCCESS) { p.status = -1102; return p; }
  if (set_batch(p.a, key.batch, key.sb) || set_batch(p.b, key.batch, key.sa) || set_batch(p.c, key.batch, key.sc)) { p.status = -1103; return p; }
  hipblasLtMatmulPreference_t pref = nullptr;
  if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) { p.status = -1104; return p; }
  size_t ws = kWorkspaceBytes;
  hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws, sizeof(ws));
  // the default: the heuristic's single best (exactly what at::mm would be given); it is also the reference the other candidates
  // are validated against.  fp32 plans keep it: the ranked list for fp32 holds kernels that are 2e-3 off (seen on a
  // [8200 x 1032] x [1032 x 64] product), and the MSDeformAttn projections are fp32 on purpose (deformable_transformer.py:250).
  hipblasLtMatmulHeuristicResult_t res[kRanked];
  int found = 0;
  hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(s.handle, p.desc, p.a, p.b, p.c, p.c, pref, 1, res, &found);
  if (st != HIPBLAS_STATUS_SUCCESS || found < 1) { hipblasLtMatmulPreferenceDestroy(pref); p.status = -1105; return p; }
  p.algo = res[0].algo;
  p.workspace = res[0].workspaceSize;
  p.cand[0] = p.algo;
  p.cand_ws[0] = p.workspace;
  p.ncand = 1;
  // hipBLASLt's hand-written "Custom_Cijk_..._MT256x256x64" kernels returned intermittently WRONG rows on a [8200 x 64] x [64 x 1032]
  // bf16 product (K = one depth-64 iteration; tools/dbg_brd_after_mso.py with OCPG_GEMM_TUNE_LOG=1: the same kernel agrees with the
  // default in one run and is off by the size of the values in the next): never a candidate below four K iterations
  // ... and at any K never a candidate the heuristic did not put FIRST itself (a kernel that is wrong one run in many passes any
  // finite number of checks; as hipBLASLt's own first choice it is what every other caller of the library runs too)
  auto custom = [&](hipblasLtMatmulAlgo_t a) { return hipblaslt_ext::getKernelNameFromAlgo(s.handle, a).rfind("Custom_", 0) == 0; };
  auto unsafe = [&](hipblasLtMatmulAlgo_t a) { return custom(a); };
  const bool default_unsafe = key.k < 256 && custom(p.algo);
  if ((tuning() && (key.dtype != 0 || tuning_fp32())) || default_unsafe) {
    found = 0;
    st = hipblasLtMatmulAlgoGetHeuristic(s.handle, p.desc, p.a, p.b, p.c, p.c, pref, kRanked - 1, res, &found);
    if (default_unsafe) p.ncand = 0;            // the first safe kernel of the ranked list becomes the default
    for (int i = 0; st == HIPBLAS_STATUS_SUCCESS && i < found && p.ncand < kCandidates; ++i) {
      if (res[i].state != HIPBLAS_STATUS_SUCCESS || res[i].workspaceSize > kWorkspaceBytes || unsafe(res[i].algo)) continue;
      if (p.ncand == 0) { p.algo = res[i].algo; p.workspace = res[i].workspaceSize; }
      if (default_unsafe && !(tuning() && (key.dtype != 0 || tuning_fp32())) && p.ncand == 1) break;
      p.cand[p.ncand] = res[i].algo;
      p.cand_ws[p.ncand++] = res[i].workspaceSize;
    }
  }
  // OCPG_GEMM_TUNE_ALL (default 1, round 4): the heuristic ranks by a model; time EVERY Tensile kernel that supports the problem
  // (hipblaslt_ext::getAllAlgos + matmulIsAlgoSupported, same validation against the default as the ranked ones).  The list order is
  // the library's own, so the candidate indices stay comparable between ranks (gemm_sync).
  // Not for products of fewer than four depth-64 K iterations: both times a validated kernel was later seen wrong (round 3's Custom_
  // kernels, round 4's fp32 candidate) the product was tests' [8200 x 64] x [64 x 1032] -- short-K problems keep the ranked list.
  if (tune_all() && tuning() && (key.dtype != 0 || tuning_fp32()) && p.ncand >= 1 && key.k >= 256) {
    std::vector<hipblasLtMatmulHeuristicResult_t> all;
    if (hipblaslt_ext::getAllAlgos(s.handle, hipblaslt_ext::GemmType::HIPBLASLT_GEMM, opa, opb, in, in, out, out, HIPBLAS_COMPUTE_32F, all) ==
        HIPBLAS_STATUS_SUCCESS) {
      const float one = 1.f, zero = 0.f;
      for (size_t i = 0; i < all.size() && p.ncand < kCandidates; ++i) {
        size_t need = 0;
        if (hipblaslt_ext::matmulIsAlgoSupported(s.handle, p.desc, &one, p.a, p.b, &zero, p.c, p.c, all[i].algo, need) != HIPBLAS_STATUS_SUCCESS) continue;
        if (need > kWorkspaceBytes || unsafe(all[i].algo)) continue;
        const int idx = hipblaslt_ext::getIndexFromAlgo(all[i].algo);
        bool dup = false;
        for (int j = 0; j < p.ncand && !dup; ++j) dup = hipblaslt_ext::getIndexFromAlgo(p.cand[j]) == idx;
        if (dup) continue;
        p.cand[p.ncand] = all[i].algo;
        p.cand_ws[p.ncand++] = need;
      }
    }
  }
  hipblasLtMatmulPreferenceDestroy(pref);
  return p;
}

__device__ __forceinline__ float cmp_ld(const void* p, int dt, long long i) {
  if (dt == 0) return reinterpret_cast<const float*>(p)[i];
  if (dt == 2) return __half2float(reinterpret_cast<const __half*>(p)[i]);
  return __bfloat162float(reinterpret_cast<const __hip_bfloat16*>(p)[i]);
}
// out[0] = max |c - ref|, out[1] = max |ref| over n elements (non-negative floats order like their bit patterns; NaN counts as +inf)
__global__ void k_cmp(const void* c, const void* ref, int dt, long long n, unsigned* out) {
  float d = 0.f, r = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float a = cmp_ld(c, dt, i), b = cmp_ld(ref, dt, i);
    float e = fabsf(a - b);
    if (!(e <= 3.0e38f)) e = __uint_as_float(0x7f800000u);
    d = fmaxf(d, e);
    r = fmaxf(r, fabsf(b));
  }
  for (int o = 32; o > 0; o >>= 1) {
    d = fmaxf(d, __shfl_xor(d, o, 64));
    r = fmaxf(r, __shfl_xor(r, o, 64));
  }
  if ((threadIdx.x & 63) == 0) {
    atomicMax(out, __float_as_uint(d));
    atomicMax(out + 1, __float_as_uint(r));
  }
}

// second pass: number of elements with |c - ref| > rtol * |ref| + atol_frac * max|ref| (max|ref| = out[1] of the first pass, read on
// the device: no host round trip between the passes) -> out[2].  The global bound alone (max|diff| against max|ref|) lets
// small-magnitude elements be arbitrarily wrong in relative terms.
__global__ void k_cmp_count(const void* c, const void* ref, int dt, long long n, float rtol, float atol_frac, unsigned* out) {
  const float atol = atol_frac * __uint_as_float(out[1]);
  unsigned bad = 0;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float a = cmp_ld(c, dt, i), b = cmp_ld(ref, dt, i);
    bad += !(fabsf(a - b) <= rtol * fabsf(b) + atol);
  }
  for (int o = 32; o > 0; o >>= 1) bad += __shfl_xor(bad, o, 64);
  if ((threadIdx.x & 63) == 0 && bad) atomicAdd(out + 2, bad);
}

// First use of a plan whose call is repeatable (it does not accumulate into its own output): run every candidate of the heuristic's
// ranked list on the caller's operands and keep the fastest ONE WHOSE RESULT AGREES with the default's (candidate 0: hipBLASLt's own
// single choice, what at::mm would run) to the rounding of the output type.  Only bf16 / fp16 plans carry more than the default.  hipBLASLt's first choice is a
// model's guess; on this path's shapes (tall-skinny token matrices against 64..2048 channels) a later candidate is often faster.
// The shapes repeat every step, so the cost (a few launches and one event wait per candidate, once per plan) is paid in the first
// step only.  Never inside a stream capture (the plan stays untuned until an eager call sees it).
// `launch(algo, workspace_bytes)` issues the matmul into C (out_dtype code dt; `span` elements from C cover the output).
template <typename Launch> void tune(State& s, Plan& p, hipStream_t st, void* workspace, const void* C, int dt, long long span, Launch&& launch) {
  if (p.tuned) return;
  if (p.ncand <= 1 || !tuning()) { p.tuned = true; return; }
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
  const long long n = span;                   // the WHOLE output is compared: a candidate that is wrong only in its edge tiles was seen
  const size_t esize = dt == 0 ? 4 : 2;
  bool ready = s.ev0 || (hipEventCreate(&s.ev0) == hipSuccess && hipEventCreate(&s.ev1) == hipSuccess &&
                         hipMalloc(reinterpret_cast<void**>(&s.cmp_out), 16) == hipSuccess);
  if (ready && s.cmp_bytes < (size_t)n * esize) {
    if (s.cmp_ref) (void)hipFree(s.cmp_ref);
    s.cmp_ref = nullptr;
    s.cmp_bytes = 0;
    if (hipMalloc(&s.cmp_ref, (size_t)n * esize) == hipSuccess) s.cmp_bytes = (size_t)n * esize;
    else ready = false;
  }
  if (!ready) {
    p.tuned = true;
    (void)hipGetLastError();
    return;
  }
  const float tol = dt == 0 ? 1e-5f : dt == 1 ? 8e-3f : 1e-3f;
  int best = -1, ref = -1;
  float best_ms = 0.f;
  static const bool log = [] { const char* e = getenv("OCPG_GEMM_TUNE_LOG"); return e && e[0] == '1'; }();
  for (int i = 0; i < p.ncand; ++i) {
    // every candidate starts from a zeroed workspace: kernels that keep flags / partial tiles there (split-K, stream-K) were seen to
    // return wrong rows when they ran on what ANOTHER algorithm had left behind
    if (p.cand_ws[i] && ocpg_fill::zero_async(workspace, p.cand_ws[i], st) != hipSuccess) continue;
    if (launch(p.cand[i], p.cand_ws[i]) != HIPBLAS_STATUS_SUCCESS) continue;          // also the warm-up run
    if (ref < 0) {
      if (hipMemcpyAsync(s.cmp_ref, C, (size_t)n * esize, hipMemcpyDeviceToDevice, st) != hipSuccess) break;
      ref = i;
    } else {
      (void)ocpg_fill::zero_async(s.cmp_out, 16, st);
      hipLaunchKernelGGL(k_cmp, dim3(256), dim3(256), 0, st, C, (const void*)s.cmp_ref, dt, n, s.cmp_out);
      // per element: a few roundings of the output type relative to the element itself, plus a floor of tol / 4 of the largest
      hipLaunchKernelGGL(k_cmp_count, dim3(256), dim3(256), 0, st, C, (const void*)s.cmp_ref, dt, n, 4.f * tol, 0.25f * tol, s.cmp_out);
    }
    (void)hipEventRecord(s.ev0, st);
    bool ok = true;
    for (int r = 0; r < 3 && ok; ++r) ok = launch(p.cand[i], p.cand_ws[i]) == HIPBLAS_STATUS_SUCCESS;
    (void)hipEventRecord(s.ev1, st);
    if (i != ref) {     // ... and the result of the LAST timed run too (an intermittently wrong kernel has to be right every time)
      hipLaunchKernelGGL(k_cmp, dim3(256), dim3(256), 0, st, C, (const void*)s.cmp_ref, dt, n, s.cmp_out);
      hipLaunchKernelGGL(k_cmp_count, dim3(256), dim3(256), 0, st, C, (const void*)s.cmp_ref, dt, n, 4.f * tol, 0.25f * tol, s.cmp_out);
      (void)hipStreamSynchronize(st);
    }
    float ms = 0.f;
    if (hipEventSynchronize(s.ev1) != hipSuccess || hipEventElapsedTime(&ms, s.ev0, s.ev1) != hipSuccess || !ok) continue;
    if (i != ref) {
      unsigned bits[3] = {0x7f800000u, 0u, 1u};
      if (hipMemcpy(bits, s.cmp_out, 12, hipMemcpyDeviceToHost) != hipSuccess) continue;
      float diff, mag;
      memcpy(&diff, &bits[0], 4);
      memcpy(&mag, &bits[1], 4);
      if (log) fprintf(stderr, "[ocpg_gemm tune] cand %d ws %zu: %.1f us  max|diff| %.3g of %.3g, %u elements out of bound  %s\n", i, p.cand_ws[i],
                       ms * 1000.f / 3.f, diff, mag, bits[2], hipblaslt_ext::getKernelNameFromAlgo(s.handle, p.cand[i]).c_str());
      if (!(diff <= tol * mag + 1e-30f) || bits[2] != 0u) { s.tuned_rejected += 1; continue; }
    } else if (log) {
      fprintf(stderr, "[ocpg_gemm tune] cand %d ws %zu: %.1f us  (reference)  %s\n", i, p.cand_ws[i], ms * 1000.f / 3.f,
              hipblaslt_ext::getKernelNameFromAlgo(s.handle, p.cand[i]).c_str());
    }
    if (best < 0 || ms < best_ms) { best = i; best_ms = ms; }
  }
  // the winner once more, now on the workspace as the other candidates left it (the state it will meet between other plans' calls):
  // it has to reproduce the reference again, or the default stays
  if (best >= 0 && best != ref && ref >= 0) {
    bool ok = launch(p.cand[best], p.cand_ws[best]) == HIPBLAS_STATUS_SUCCESS;
    unsigned bits[3] = {0x7f800000u, 0u, 1u};
    if (ok) {
      ok = ocpg_fill::zero_async(s.cmp_out, 16, st) == hipSuccess;
      hipLaunchKernelGGL(k_cmp, dim3(256), dim3(256), 0, st, C, (const void*)s.cmp_ref, dt, n, s.cmp_out);
      hipLaunchKernelGGL(k_cmp_count, dim3(256), dim3(256), 0, st, C, (const void*)s.cmp_ref, dt, n, 4.f * tol, 0.25f * tol, s.cmp_out);
      ok = ok && hipStreamSynchronize(st) == hipSuccess && hipMemcpy(bits, s.cmp_out, 12, hipMemcpyDeviceToHost) == hipSuccess;
    }
    float diff, mag;
    memcpy(&diff, &bits[0], 4);
    memcpy(&mag, &bits[1], 4);
    if (log) fprintf(stderr, "[ocpg_gemm tune] winner %d re-run: max|diff| %.3g of %.3g, %u elements out of bound\n", best, diff, mag, bits[2]);
    if (!ok || !(diff <= tol * mag + 1e-30f) || bits[2] != 0u) {
      s.tuned_rejected += 1;
      best = ref;
    }
  }
  if (workspace && p.ncand > 1) (void)ocpg_fill::zero_async(workspace, kWorkspaceBytes, st);
  if (best >= 0) {
    p.algo = p.cand[best];
    p.workspace = p.cand_ws[best];
    p.picked = best;
    s.tuned_plans += 1;
    s.tuned_changed += best != 0;
  }
  (void)hipGetLastError();
  p.tuned = true;
}

}  // namespace

extern "C" int ocpg_gemm(const void* A, const void* B, void* C, const void* bias, int dtype, int out_dtype, int transA, int transB,
                         long long M, long long N, long long K, long long lda, long long ldb, long long ldc, long long batch,
                         long long strideA, long long strideB, long long strideC, float alpha, float beta, void* stream) {
  if (dtype < 0 || dtype > 2 || out_dtype < 0 || out_dtype > 2) return -1010;
  if (M < 0 || N < 0 || K < 0 || batch < 1) return -1006;
  if (M == 0 || N == 0) return 0;
  if (K == 0) return -1007;                      // caller handles the empty contraction (C = beta*C + bias)
  if (!A) return -1001;
  if (!B) return -1002;
  if (!C) return -1003;
  State* sp = state();
  if (!sp) return -1098;
  State& s = *sp;
  std::lock_guard<std::mutex> lock(s.mu);
  void* workspace = nullptr;
  if (int e = prepare(s, (hipStream_t)stream, &workspace)) return e;
  const Key key{dtype, out_dtype, transA != 0, transB != 0, bias != nullptr, beta == 0.f, 0, M, N, K, lda, ldb, ldc, batch,
                batch > 1 ? strideA : 0, batch > 1 ? strideB : 0, batch > 1 ? strideC : 0};
  Plan& p = plan_for(s, key);
  if (p.status) return p.status;
  if (bias) hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &bias, sizeof(bias));
  if (!p.tuned) {
    if (beta == 0.f)
      tune(s, p, (hipStream_t)stream, workspace, C, out_dtype, (batch - 1) * (batch > 1 ? strideC : 0) + (M - 1) * ldc + N,
           [&](const hipblasLtMatmulAlgo_t& algo, size_t ws) {
             return hipblasLtMatmul(s.handle, p.desc, &alpha, B, p.a, A, p.b, &beta, C, p.c, C, p.c, &algo, workspace, ws, (hipStream_t)stream);
           });
    else {
      // C += ... cannot be repeated on the caller's C: the candidates are timed and compared with beta = 0 on a scratch output of the
      // same layout (same kernels, same operands), and the winner serves the accumulating calls
      hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
      const size_t span = (size_t)((batch - 1) * (batch > 1 ? strideC : 0) + (M - 1) * ldc + N), bytes = span * (out_dtype == 0 ? 4 : 2);
      if (p.ncand > 1 && tuning() && hipStreamIsCapturing((hipStream_t)stream, &cs) == hipSuccess && cs == hipStreamCaptureStatusNone) {
        if (s.scratch_bytes < bytes) {
          if (s.scratch_c) (void)hipFree(s.scratch_c);
          s.scratch_c = nullptr;
          s.scratch_bytes = 0;
          if (hipMalloc(&s.scratch_c, bytes) == hipSuccess) s.scratch_bytes = bytes;
        }
        if (s.scratch_c) {
          const float zero = 0.f;
          void* Cs = s.scratch_c;
          tune(s, p, (hipStream_t)stream, workspace, Cs, out_dtype, (long long)span, [&](const hipblasLtMatmulAlgo_t& algo, size_t ws) {
            return hipblasLtMatmul(s.handle, p.desc, &alpha, B, p.a, A, p.b, &zero, Cs, p.c, Cs, p.c, &algo, workspace, ws, (hipStream_t)stream);
          });
        } else {
          (void)hipGetLastError();
          p.tuned = true;
        }
      } else if (cs == hipStreamCaptureStatusNone) {
        p.tuned = true;
      }
    }
  }
  const hipblasStatus_t st = hipblasLtMatmul(s.handle, p.desc, &alpha, B, p.a, A, p.b, &beta, C, p.c, C, p.c, &p.algo, workspace,
                                             p.workspace, (hipStream_t)stream);
  return st == HIPBLAS_STATUS_SUCCESS ? 0 : -1200 - (int)st;
}

// D[M,N] = act(scale[n] * (A W^T)[m,n] + shift[n] (+ skip[m,n]))  -- 1x1 conv + frozen-BN affine (+ residual) (+ ReLU) in the GEMM's
// epilogue: A [M,K] (lda = K), W [N,K], skip / D [M,N] dense; scale, shift fp32 [N].  Returns -1105 when hipBLASLt has no
// kernel for that combination (the caller then runs GEMM + bn_act).
extern "C" int ocpg_gemm_bn_act(const void* A, const void* W, void* D, const float* scale, const float* shift, const void* skip, int relu,
                                int dtype, long long M, long long N, long long K, void* stream) {
  if (dtype < 0 || dtype > 2) return -1010;
  if (M <= 0 || N <= 0 || K <= 0) return -1006;
  if (!A) return -1001;
  if (!W) return -1002;
  if (!D) return -1003;
  if (!scale || !shift) return -1004;
  State* sp = state();
  if (!sp) return -1098;
  State& s = *sp;
  std::lock_guard<std::mutex> lock(s.mu);
  void* workspace = nullptr;
  if (int e = prepare(s, (hipStream_t)stream, &workspace)) return e;
  const Key key{dtype, dtype, 0, 1, 1, skip == nullptr, (relu ? 1 : 0) | 2 | 4, M, N, K, K, K, N, 1, 0, 0, 0};
  Plan& p = plan_for(s, key);
  if (p.status) return p.status;
  hipblasLtMatmulDescSetAttribute(p.desc, HIPBLASLT_MATMUL_DESC_BIAS_POINTER, &shift, sizeof(shift));
  const float beta = skip ? 1.f : 0.f;
  if (!p.tuned) {
    if (skip != D)
      tune(s, p, (hipStream_t)stream, workspace, D, dtype, M * N, [&](const hipblasLtMatmulAlgo_t& algo, size_t ws) {
        return hipblasLtMatmul(s.handle, p.desc, scale, W, p.a, A, p.b, &beta, skip ? skip : D, p.c, D, p.c, &algo, workspace, ws, (hipStream_t)stream);
      });
    else
      p.tuned = true;
  }
  const hipblasStatus_t st = hipblasLtMatmul(s.handle, p.desc, scale, W, p.a, A, p.b, &beta, skip ? skip : D, p.c, D, p.c, &p.algo,
                                             workspace, p.workspace, (hipStream_t)stream);
  return st == HIPBLAS_STATUS_SUCCESS ? 0 : -1200 - (int)st;
}

extern "C" long long ocpg_gemm_tune_rejected(void) {      // candidates of the current device dropped because their result differed
  State* sp = state();
  if (!sp) return -1;
  std::lock_guard<std::mutex> lock(sp->mu);
  return sp->tuned_rejected;
}

extern "C" long long ocpg_gemm_tuned(long long* changed) {      // plans of the current device that were timed / that left the heuristic's first choice
  State* sp = state();
  if (!sp) return -1;
  std::lock_guard<std::mutex> lock(sp->mu);
  if (changed) *changed = sp->tuned_changed;
  return sp->tuned_plans;
}

// ---- rank-consistent plan choices (multi-GPU data parallel: one rank times the candidates, every rank runs its winners) ------------------
extern "C" void ocpg_gemm_set_tuning(int on) { g_tuning_override = on; }      // 1 / 0: candidate timing on / off in this process; -1: OCPG_GEMM_TUNE decides

extern "C" long long ocpg_gemm_export_picks(long long* buf, long long cap_pairs) {      // [key hash, candidate index] of every timed plan; returns the pair count
  State* sp = state();
  if (!sp) return -1;
  std::lock_guard<std::mutex> lock(sp->mu);
  long long n = 0;
  for (auto& kv : sp->plans) {
    if (!kv.second.tuned || kv.second.ncand <= 1) continue;
    if (buf && n < cap_pairs) { buf[2 * n] = (long long)KeyHash()(kv.first); buf[2 * n + 1] = kv.second.picked; }
    ++n;
  }
  return n;
}

extern "C" int ocpg_gemm_import_picks(const long long* buf, long long n_pairs) {      // applies to the plans that exist and to those built later
  State* sp = state();
  if (!sp || (n_pairs > 0 && !buf)) return -1;
  std::lock_guard<std::mutex> lock(sp->mu);
  for (long long i = 0; i < n_pairs; ++i) sp->imported[(uint64_t)buf[2 * i]] = (int)buf[2 * i + 1];
  for (auto& kv : sp->plans) {
    auto im = sp->imported.find((uint64_t)KeyHash()(kv.first));
    if (im == sp->imported.end()) continue;
    Plan& p = kv.second;
    if (im->second >= 0 && im->second < p.ncand) { p.algo = p.cand[im->second]; p.workspace = p.cand_ws[im->second]; p.picked = im->second; }
    p.tuned = true;
  }
  return 0;
}

extern "C" long long ocpg_gemm_plans(void) {      // of the current device
  State* sp = state();
  if (!sp) return -1;
  std::lock_guard<std::mutex> lock(sp->mu);
  return (long long)sp->plans.size();
}
