"""Where does the regenerated torch.normal stream differ from torch's?  Per dtype / size: mismatch
count, first positions and values; the device properties ATen's launch policy reads; a numpy
restatement of element 0 (integer stream + float64 Box-Muller) to tell an integer-stream error
from a rounding one."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from ecoflap_amd import hip

k = hip.HipKernels()
pr = torch.cuda.get_device_properties(0)
print("SMs", pr.multi_processor_count, "maxThreadsPerSM", pr.max_threads_per_multi_processor, pr.name)


def philox(c, key):
    M0, M1, W0, W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
    c = list(c); k0, k1 = key
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [((p1 >> 32) ^ c[1] ^ k0) & 0xffffffff, p1 & 0xffffffff, ((p0 >> 32) ^ c[3] ^ k1) & 0xffffffff, p0 & 0xffffffff]
        k0, k1 = (k0 + W0) & 0xffffffff, (k1 + W1) & 0xffffffff
    return c


def bm(x, y):
    u = x * 2.0 ** -32 + 2.0 ** -32
    v = y * (2 * np.pi * 2.0 ** -32) + 2 * np.pi * 2.0 ** -32
    s = np.sqrt(-2 * np.log(u))
    return np.sin(v) * s, np.cos(v) * s


for dt in (torch.float32, torch.float16, torch.bfloat16):
    for n, seed in ((8, 1), (1003, 123456789), (4100, 5), (256 * 2048 + 3, 7), (4 * 524288 + 4104, 987654321)):
        torch.manual_seed(seed)
        want = torch.normal(mean=0, std=1, size=(n,), device="cuda", dtype=dt)
        got = torch.empty_like(want)
        k.zo_fill_normal_torch(got, seed)
        T = k.torch_normal_threads(n, want.device)
        vi = torch.int32 if dt == torch.float32 else torch.int16
        d = (want.view(vi) != got.view(vi)).nonzero().flatten()
        print(dt, "n", n, "seed", seed, "T", T, "mismatch", d.numel(), "first", d[:6].tolist())
        if d.numel():
            i = d[:4]
            print("   torch", want[i].tolist(), "\n   mine ", got[i].tolist())
            w0 = philox([0, 0, 0, 0], (seed & 0xffffffff, seed >> 32))
            print("   numpy elem0 (sin, cos):", bm(w0[0], w0[1]), " torch elem0:", float(want[0]), " mine elem0:", float(got[0]))
            w1 = philox([0, 0, 1, 0], (seed & 0xffffffff, seed >> 32))
            print("   numpy thread1 (sin):", bm(w1[0], w1[1])[0], " torch elem1:", float(want[1]) if n > 1 else None,
                  " mine elem1:", float(got[1]) if n > 1 else None)
