#!/usr/bin/env python3
"""Is the step bound by the host (Python + ctypes launch overhead) or by the device?  Prints, per step, the host time to
ENQUEUE a step (no synchronisation) next to the device time of the step, and samples the shader clock / socket power
(rocm-smi) while the steps run.  GPU only; dev tool.   env: arch=vit_b_16|vit_s_16|...   argv: steps"""
import os, sys, time, subprocess, threading, re, statistics
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench as B
from noise_robust_vit_amd.train import TrainConfig, Trainer

arch = os.environ.get("arch", "vit_b_16")
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
dev = torch.device("cuda:0")
kind, kw = B.ARCHS[arch][0], B.ARCHS[arch][1]
batch = 128 if arch.endswith("l_16") else 256
model = B.build_model(arch).to(dev).train()
trainer = Trainer(model, TrainConfig(lr=5e-4, weight_decay=0.05, grad_max_norm=5.0), None,
                  compute_loss=(lambda m, xb, yb: m(xb)) if kind == "mae" else None)
gen = torch.Generator(device=dev).manual_seed(1234)
x = torch.randn(batch, 3, kw["image_size"], kw["image_size"], generator=gen, device=dev).to(torch.bfloat16)
y = torch.randint(0, 1000, (batch,), generator=gen, device=dev)
for _ in range(5): trainer.step(x, y)
torch.cuda.synchronize()

samples, stop = [], False
def sampler():
    while not stop:
        try:
            o = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", o)
            pw = re.search(r"Power \(W\): ([\d.]+)", o)
            samples.append((int(sclk.group(1)) if sclk else -1, float(pw.group(1)) if pw else -1.0))
        except Exception as e:
            samples.append((-1, -1.0))
        time.sleep(0.05)
th = threading.Thread(target=sampler, daemon=True); th.start()

# (1) host enqueue time with the device kept busy (queue never empty): enqueue `steps` steps back to back
t0 = time.perf_counter()
for _ in range(steps): trainer.step(x, y)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"{arch}: host enqueue {1e3 * (t1 - t0) / steps:7.3f} ms/step, wall incl. final sync {1e3 * (t2 - t0) / steps:7.3f} ms/step "
      f"(queue drained {1e3 * (t2 - t1):.1f} ms after the last enqueue)")
# (2) longer run for the clock / power samples
for _ in range(3 * steps): trainer.step(x, y)
torch.cuda.synchronize()
stop = True; th.join()
ok = [s for s in samples if s[0] > 0]
if ok:
    print(f"rocm-smi during the steps ({len(ok)} samples): sclk median {statistics.median(s[0] for s in ok)} MHz "
          f"(min {min(s[0] for s in ok)}, max {max(s[0] for s in ok)}), power median {statistics.median(s[1] for s in ok):.0f} W (max {max(s[1] for s in ok):.0f})")
else:
    print("rocm-smi gave no samples:", samples[:3])
