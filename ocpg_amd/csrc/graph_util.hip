// HIP-graph hygiene for whole-step capture (bench.py --graph).
//
// ocpg_graph_replace_memsets: every memset node of a captured graph is replaced by a kernel node running ocpg_fill::k_fill with
// the same destination / value / extent and the same edges.  Reason (measured, tools/dbg_graph_memset.py): with the HIP runtime
// that ships with torch 2.10+rocm7.0 a captured hipMemsetAsync zeroes correctly on the FIRST launch of the instantiated graph
// and writes a stale 16-byte pattern (two host pointers) on every later launch.  torch's multi-block reductions clear their
// semaphores with hipMemsetAsync, so from the second replay on those reductions never see "last block" and return garbage --
// the "replays go non-finite" hazard recorded in round 1.  Kernel nodes replay correctly, so the graph is repaired before it
// is instantiated.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "../../include/ocpg_hip.h"
#include "fill.h"

extern "C" int ocpg_graph_replace_memsets(void* graph_, int* n_replaced) {
  hipGraph_t graph = (hipGraph_t)graph_;
  if (!graph) return -1001;
  size_t n = 0;
  hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
  if (e != hipSuccess) return -(int)e;
  std::vector<hipGraphNode_t> nodes(n);
  if (n) {
    e = hipGraphGetNodes(graph, nodes.data(), &n);
    if (e != hipSuccess) return -(int)e;
  }
  int replaced = 0;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType type;
    e = hipGraphNodeGetType(nodes[i], &type);
    if (e != hipSuccess) return -(int)e;
    if (type != hipGraphNodeTypeMemset) continue;
    hipMemsetParams mp;
    e = hipGraphMemsetNodeGetParams(nodes[i], &mp);
    if (e != hipSuccess) return -(int)e;
    size_t nd = 0, nout = 0;
    e = hipGraphNodeGetDependencies(nodes[i], nullptr, &nd);
    if (e != hipSuccess) return -(int)e;
    std::vector<hipGraphNode_t> deps(nd);
    if (nd && (e = hipGraphNodeGetDependencies(nodes[i], deps.data(), &nd)) != hipSuccess) return -(int)e;
    e = hipGraphNodeGetDependentNodes(nodes[i], nullptr, &nout);
    if (e != hipSuccess) return -(int)e;
    std::vector<hipGraphNode_t> outs(nout);
    if (nout && (e = hipGraphNodeGetDependentNodes(nodes[i], outs.data(), &nout)) != hipSuccess) return -(int)e;

    unsigned char* dst = (unsigned char*)mp.dst;
    unsigned value = mp.value;
    int elem = (int)mp.elementSize;
    size_t row_bytes = mp.width * mp.elementSize, rows = mp.height ? mp.height : 1, pitch = mp.pitch;
    if (elem != 1 && elem != 2 && elem != 4) return -2000;
    void* args[] = {&dst, &value, &elem, &row_bytes, &rows, &pitch};
    hipKernelNodeParams kp = {};
    kp.func = (void*)ocpg_fill::k_fill;
    kp.gridDim = dim3(ocpg_fill::fill_blocks(row_bytes));
    kp.blockDim = dim3(256);
    kp.sharedMemBytes = 0;
    kp.kernelParams = args;
    kp.extra = nullptr;
    hipGraphNode_t knode;
    e = hipGraphAddKernelNode(&knode, graph, deps.data(), nd, &kp);
    if (e != hipSuccess) return -(int)e;
    if (nout) {
      std::vector<hipGraphNode_t> from(nout, knode);
      e = hipGraphAddDependencies(graph, from.data(), outs.data(), nout);
      if (e != hipSuccess) return -(int)e;
    }
    e = hipGraphDestroyNode(nodes[i]);
    if (e != hipSuccess) return -(int)e;
    ++replaced;
  }
  if (n_replaced) *n_replaced = replaced;
  return 0;
}

// Topology of a captured graph: out[0..8] = nodes, edges, roots, max out-degree, max in-degree, kernel / memset / memcpy / other
// node counts.  A single-stream capture must be one chain (roots 1, degrees <= 1); anything else means a library forked streams.
extern "C" int ocpg_graph_stats(void* graph_, long long* out) {
  hipGraph_t graph = (hipGraph_t)graph_;
  if (!graph || !out) return -1001;
  size_t n = 0, ne = 0, nr = 0;
  hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
  if (e != hipSuccess) return -(int)e;
  std::vector<hipGraphNode_t> nodes(n);
  if (n && (e = hipGraphGetNodes(graph, nodes.data(), &n)) != hipSuccess) return -(int)e;
  if ((e = hipGraphGetEdges(graph, nullptr, nullptr, &ne)) != hipSuccess) return -(int)e;
  if ((e = hipGraphGetRootNodes(graph, nullptr, &nr)) != hipSuccess) return -(int)e;
  long long max_out = 0, max_in = 0, nk = 0, nm = 0, nc = 0, no = 0;
  for (size_t i = 0; i < n; ++i) {
    size_t a = 0, b = 0;
    if ((e = hipGraphNodeGetDependentNodes(nodes[i], nullptr, &a)) != hipSuccess) return -(int)e;
    if ((e = hipGraphNodeGetDependencies(nodes[i], nullptr, &b)) != hipSuccess) return -(int)e;
    max_out = std::max(max_out, (long long)a);
    max_in = std::max(max_in, (long long)b);
    hipGraphNodeType t;
    if ((e = hipGraphNodeGetType(nodes[i], &t)) != hipSuccess) return -(int)e;
    if (t == hipGraphNodeTypeKernel) ++nk;
    else if (t == hipGraphNodeTypeMemset) ++nm;
    else if (t == hipGraphNodeTypeMemcpy) ++nc;
    else ++no;
  }
  out[0] = (long long)n; out[1] = (long long)ne; out[2] = (long long)nr; out[3] = max_out; out[4] = max_in;
  out[5] = nk; out[6] = nm; out[7] = nc; out[8] = no;
  return 0;
}

// Memcpy nodes of a captured graph: out[4 * i ..] = kind (hipMemcpyKind), bytes, source pointer, destination pointer; returns the
// number of memcpy nodes (at most cap are written) or a negative error.  A host-to-device node re-reads its HOST source on every
// replay: the source must stay alive and unchanged, or the replay uploads garbage.
extern "C" int ocpg_graph_memcpy_nodes(void* graph_, long long* out, int cap) {
  hipGraph_t graph = (hipGraph_t)graph_;
  if (!graph || !out) return -1001;
  size_t n = 0;
  hipError_t e = hipGraphGetNodes(graph, nullptr, &n);
  if (e != hipSuccess) return -(int)e;
  std::vector<hipGraphNode_t> nodes(n);
  if (n && (e = hipGraphGetNodes(graph, nodes.data(), &n)) != hipSuccess) return -(int)e;
  int k = 0;
  for (size_t i = 0; i < n; ++i) {
    hipGraphNodeType t;
    if ((e = hipGraphNodeGetType(nodes[i], &t)) != hipSuccess) return -(int)e;
    if (t != hipGraphNodeTypeMemcpy) continue;
    hipMemcpy3DParms p;
    if ((e = hipGraphMemcpyNodeGetParams(nodes[i], &p)) != hipSuccess) return -(int)e;
    if (k < cap) {
      out[4 * k + 0] = (long long)p.kind;
      out[4 * k + 1] = (long long)(p.extent.width * std::max<size_t>(p.extent.height, 1) * std::max<size_t>(p.extent.depth, 1));
      out[4 * k + 2] = (long long)(uintptr_t)p.srcPtr.ptr;
      out[4 * k + 3] = (long long)(uintptr_t)p.dstPtr.ptr;
    }
    ++k;
  }
  return k;
}
