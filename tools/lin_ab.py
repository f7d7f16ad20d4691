import sys, torch, statistics
sys.path.insert(0, ".")
from camera_linearity_amd import engine
from camera_linearity_amd.synthetic import synthetic_stack_device, synthetic_icrf
dev = torch.device("cuda:0")
frames, stds, t = synthetic_stack_device(7, 2, 4096, 4096, device=dev, with_std=True)
icrf, diff = synthetic_icrf(); icrf = torch.as_tensor(icrf, device=dev); diff = torch.as_tensor(diff, device=dev)
dn = frames[1]
def span(fn, it=20):
    fn(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / it
for _ in range(200): engine.u8_to_unit(dn)
v = engine.u8_to_unit(dn)
r = {"u8_to_unit": [], "linearize": [], "linearize_std": [], "linearize_f64": [], "linearize_f64_std": [], "linearize_1d_lut": []}
lut1 = icrf[:, 0].contiguous()
for _ in range(5):
    r["u8_to_unit"].append(span(lambda: engine.u8_to_unit(dn)))
    r["linearize"].append(span(lambda: engine.linearize(dn, None, icrf)))
    r["linearize_std"].append(span(lambda: engine.linearize(dn, stds[1], icrf, diff)))
    r["linearize_f64"].append(span(lambda: engine.linearize(v, None, icrf)))
    r["linearize_f64_std"].append(span(lambda: engine.linearize(v, stds[1], icrf, diff)))
    r["linearize_1d_lut"].append(span(lambda: engine.linearize(dn, None, lut1)))
print({k: round(statistics.median(v), 1) for k, v in r.items()})
