"""The dueling network on a 16-row tile with v_mfma_f32_16x16x32_f16 (half16.hip) against rl_policy_forward (the canonical 32-row tiles): Q values bit for bit.
   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -shared -Ireinlife_amd/csrc -Iinclude -o tools/half16/libhalf16.so tools/half16/half16.hip && python tools/half16/check.py   (GPU)"""
import ctypes as C, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench
from reinlife_amd import _lib
from reinlife_amd.worlds import pack_brain_weights
lib = _lib.lib()
h = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhalf16.so"))
hip = C.CDLL("libamdhip64.so")
w = bench.brain_weights("PERD3QN", 101)
packed = pack_brain_weights(_lib.PERD3QN, w, "cuda:0")
rng = np.random.RandomState(3)
bad = 0
for trial in range(20):
    n = 16 if trial % 3 else rng.randint(1, 17)
    scale = 10.0 ** rng.randint(-3, 3)
    obs = (rng.choice([-1, 0, 0.5, 1, 0.37, -0.81], size=(n, 153)) * scale).astype(np.float32)
    if trial % 2: obs = (rng.randn(n, 153) * scale).astype(np.float32)
    o = torch.from_numpy(obs).cuda()
    ref = torch.zeros(n, 8, device="cuda"); got = torch.zeros(n, 8, device="cuda")
    _lib.check(lib.rl_policy_forward(_lib.PERD3QN, C.c_void_p(packed.data_ptr()), C.c_void_p(o.data_ptr()), n, C.c_void_p(ref.data_ptr()), None), "fwd")
    torch.cuda.synchronize()
    # launch half16_forward<<<1, 64>>> through the HIP module API of the loaded .so: simplest is hipLaunchKernel on the symbol
    fn = C.c_void_p.in_dll(h, "half16_forward") if False else None
    args = (C.c_void_p * 4)(C.cast(C.pointer(C.c_void_p(packed.data_ptr())), C.c_void_p), C.cast(C.pointer(C.c_void_p(o.data_ptr())), C.c_void_p),
                            C.cast(C.pointer(C.c_int(n)), C.c_void_p), C.cast(C.pointer(C.c_void_p(got.data_ptr())), C.c_void_p))
    class dim3(C.Structure): _fields_ = [("x", C.c_uint), ("y", C.c_uint), ("z", C.c_uint)]
    hip.hipLaunchKernel.argtypes = [C.c_void_p, dim3, dim3, C.c_void_p, C.c_size_t, C.c_void_p]
    rc = hip.hipLaunchKernel(C.cast(h.half16_forward, C.c_void_p), dim3(1, 1, 1), dim3(64, 1, 1), args, 0, None)
    assert rc == 0, rc
    torch.cuda.synchronize()
    a, b = ref.cpu().numpy(), got.cpu().numpy()
    same = np.array_equal(a.view(np.uint32), b.view(np.uint32))
    bad += not same
    print("trial %2d n=%2d scale=%g: %s  max|d|=%.3g  ref[0]=%s" % (trial, n, scale, "BIT-IDENTICAL" if same else "DIFFERS", np.abs(a - b).max(), a[0, :3]))
print("RESULT:", "all identical" if not bad else "%d trials differ" % bad)
