#!/bin/bash
# Ablations of the pipelined attention loop (run on the GPU box): builds attention.hip stand-alone with -DATTN_ABLATE=k and times
# the backbone's self-attention shape.  0 full | 1 no exponentials | 2 no bf16 conversion + row sums | 3 no row sums | 4 no maximum
# | 5 no tile DMA, no barrier | 6 fragments not re-read from LDS | 7 DMA but no vmcnt/barrier.  (results are wrong by construction)
R=${GRAFT_REPO_ROOT:-/root/repo}
cd $R
cat > /tmp/attn_stub.cpp <<'CPP'
namespace sculpt { int num_cus() { return 256; } void set_error(const char *, ...) {} }
CPP
for k in ${ABLATE:-0 1 2 3 4 5 6 7}; do
  hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -w -fno-slp-vectorize -DATTN_ABLATE=$k -shared sculptmate_amd/csrc/attention.hip sculptmate_amd/csrc/attention_pipe.hip /tmp/attn_stub.cpp -o /tmp/libattn_$k.so || exit 1
done
python3 - <<'PY'
import ctypes, os, torch
dev = torch.device("cuda:0"); BF = torch.bfloat16
Tq = Tk = 3072; heads = 16; D = 1024
q = (torch.randn(Tq, D, device=dev) * 0.18).to(BF); k = torch.randn(Tk, D, device=dev).to(BF)
vt = torch.randn(D, Tk, device=dev).to(BF); o = torch.empty(Tq, D, dtype=BF, device=dev)
for kk in os.environ.get("ABLATE", "0 1 2 3 4 5 6 7").split():
    lib = ctypes.CDLL("/tmp/libattn_%s.so" % kk)
    f = lib.sculpt_attention_bf16_prescaled
    f.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: f(q.data_ptr(), D, k.data_ptr(), D, vt.data_ptr(), Tk, o.data_ptr(), D, Tq, Tk, heads, st)
    for _ in range(20): call()
    ts = []
    for _ in range(20):
        a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(20): call()
        b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) / 20 * 1e3)
    ts.sort(); print("ablate %s: %.1f us" % (kk, ts[len(ts) // 2]), flush=True)
PY
