"""Ad-hoc: does torch see the GPU when the library initialised HIP first?"""
import sys
sys.path.insert(0, ".")
order = sys.argv[1]
def maps():
    return sorted({l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l or "libhsa-runtime" in l})
if order == "lib-first":
    from pathtracer_amd import api
    c = api.Context(0)
    print("after lib:", maps())
    import torch
    print("torch sees", torch.cuda.device_count(), torch.cuda.is_available())
    print("after torch:", maps())
    x = torch.zeros(4, device="cuda:0"); print(x.sum().item())
else:
    import torch
    x = torch.zeros(4, device="cuda:0"); print(x.sum().item())
    print("after torch:", maps())
    from pathtracer_amd import api
    c = api.Context(0)
    print("after lib:", maps())
