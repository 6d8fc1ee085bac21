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
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <stdint.h>

#include <mutex>
#include <unordered_map>

#include "../../include/ocpg_hip.h"

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
struct Plan {
  hipblasLtMatmulDesc_t desc = nullptr;
  hipblasLtMatrixLayout_t a = nullptr, b = nullptr, c = nullptr;
  hipblasLtMatmulAlgo_t algo;
  size_t workspace = 0;
  int status = 0;
};

constexpr size_t kWorkspaceBytes = 64u << 20;
constexpr size_t kMaxPlans = 4096;        // variable-resolution training adds a plan per distinct M per layer: bounded
constexpr int kMaxDevices = 64;

// One state PER DEVICE (handle, plans) and one workspace per (device, stream): a process that drives a second GPU, or
// issues GEMMs on two streams, never shares a workspace or hands a pointer of another device to hipBLASLt.
struct State {
  std::mutex mu;
  hipblasLtHandle_t handle = nullptr;
  std::unordered_map<hipStream_t, void*> workspaces;
  std::unordered_map<Key, Plan, KeyHash> plans;
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

Plan& plan_for(State& s, const Key& key) {
  auto it = s.plans.find(key);
  if (it == s.plans.end()) {
    if (s.plans.size() >= kMaxPlans) {          // simplest bounded policy: start over (plans are rebuilt on demand)
      for (auto& kv : s.plans) destroy(kv.second);
      s.plans.clear();
    }
    it = s.plans.emplace(key, build(s, key)).first;
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
  hipblasLtMatmulHeuristicResult_t res[1];
  int found = 0;
  const hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(s.handle, p.desc, p.a, p.b, p.c, p.c, pref, 1, res, &found);
  hipblasLtMatmulPreferenceDestroy(pref);
  if (st != HIPBLAS_STATUS_SUCCESS || found < 1) { p.status = -1105; return p; }
  p.algo = res[0].algo;
  p.workspace = res[0].workspaceSize;
  return p;
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
  const hipblasStatus_t st = hipblasLtMatmul(s.handle, p.desc, scale, W, p.a, A, p.b, &beta, skip ? skip : D, p.c, D, p.c, &p.algo,
                                             workspace, p.workspace, (hipStream_t)stream);
  return st == HIPBLAS_STATUS_SUCCESS ? 0 : -1200 - (int)st;
}

extern "C" long long ocpg_gemm_plans(void) {      // of the current device
  State* sp = state();
  if (!sp) return -1;
  std::lock_guard<std::mutex> lock(sp->mu);
  return (long long)sp->plans.size();
}
