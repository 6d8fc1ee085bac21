"""Minimal reproducer hunt for the HIP-graph replay hazard: (1) a multi-block torch reduction, (2) raw hipMemsetAsync nodes of
several sizes, (3) torch.zeros / zero_ under capture.  Each case dirties the destination between replays."""
import ctypes, sys, os
import torch
dev = torch.device("cuda:0")
hip = ctypes.CDLL(os.path.join(os.path.dirname(torch.__file__), "lib", "libamdhip64.so"))
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
hip.hipMemsetAsync.restype = ctypes.c_int

def run_case(name, build, check, n=4):
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        build()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        out = build()
    res = []
    for r in range(n):
        g.replay(); torch.cuda.synchronize()
        res.append(check(out, r))
    print(f"{name:40s}", res, flush=True)

# (1) the reduction that goes wrong in the step: mean over (H, W) of a channels-last bf16 map
x = torch.randn(10, 256, 46, 78, device=dev).bfloat16().contiguous(memory_format=torch.channels_last)
want = x.float().mean(dim=(2, 3))
run_case("mean(2,3) of NHWC bf16", lambda: x.mean(dim=(2, 3)), lambda o, r: round(float((o.float() - want).abs().max()), 5))
xf = torch.randn(4096, 4096, device=dev)
wantf = float(xf.double().sum())
run_case("sum() of 16M fp32", lambda: xf.sum(), lambda o, r: round(abs(float(o) - wantf) / abs(wantf), 6))
run_case("square().mean() chain", lambda: (xf.square().mean() + 0) + xf.mean(), lambda o, r: round(float(o) - float(xf.square().mean() + xf.mean()), 6))

# (2) raw memset nodes
for nbytes in (4, 64, 256, 1024, 4096, 1 << 20):
    buf = torch.full((nbytes,), 7, dtype=torch.uint8, device=dev)
    def build(buf=buf, nbytes=nbytes):
        rc = hip.hipMemsetAsync(buf.data_ptr(), 0, nbytes, torch.cuda.current_stream().cuda_stream)
        assert rc == 0, rc
        return buf
    def check(o, r):
        bad = int((o != 0).sum())
        o.fill_(7); torch.cuda.synchronize()
        return bad
    run_case(f"hipMemsetAsync {nbytes} B", build, check)

# (3) torch-level zeroing
t = torch.full((1024,), 3.0, device=dev)
def chk(o, r):
    bad = int((o != 0).sum()); o.fill_(3.0); torch.cuda.synchronize(); return bad
run_case("t.zero_()", lambda: t.zero_(), chk)
def bz():
    z = torch.zeros(16, device=dev, dtype=torch.int32)
    return z
def chkz(o, r):
    bad = int((o != 0).sum()); o.fill_(5); torch.cuda.synchronize(); return bad
run_case("torch.zeros(16 int32) (dirtied after)", bz, chkz)

# (4) what do the surviving bytes look like?
buf = torch.full((64,), 7, dtype=torch.uint8, device=dev)
def build64():
    assert hip.hipMemsetAsync(buf.data_ptr(), 0, 64, torch.cuda.current_stream().cuda_stream) == 0
    return buf
def show(o, r):
    v = o.tolist(); o.fill_(7 + r); torch.cuda.synchronize(); return v[:32]
run_case("bytes after memset(0) of 64 B", build64, show, n=3)
buf2 = torch.full((64,), 7, dtype=torch.uint8, device=dev)
def build64b():
    assert hip.hipMemsetAsync(buf2.data_ptr(), 0xAB, 64, torch.cuda.current_stream().cuda_stream) == 0
    return buf2
run_case("bytes after memset(0xAB) of 64 B", build64b, lambda o, r: (o.tolist()[:20], o.fill_(1), torch.cuda.synchronize())[0], n=3)
# (5) torch ops that memset: zero_ on a big tensor, zeros of several sizes
for n in (1 << 10, 1 << 16, 1 << 22):
    tt = torch.full((n,), 3.0, device=dev)
    run_case(f"zero_() of {n} fp32", lambda tt=tt: tt.zero_(), lambda o, r: (int((o != 0).sum()), o.fill_(3.0), torch.cuda.synchronize())[0])
