"""Print per-kernel device times of the forward at a few batch sizes (GPU box)."""
import sys, os, json, subprocess
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
variant = os.environ.get("VARIANT", "small")
for b in (sys.argv[1:] or ["256", "2048"]):
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--batch", b, "--steps", "20",
                          "--variant", variant],
                         capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        print(out.stdout[-2000:], out.stderr[-2000:]); continue
    print(f"B={b}: {d['value']:.0f} img/s, {d['ms_per_step']:.4f} ms/step, kernel sum {d['kernel_ms_sum']:.4f}")
    print("   " + "  ".join(f"{k['kernel'].split(' ')[0]}={k['ms']*1e3:.1f}us({k['frac']*100:.1f}%)" for k in d["roofline_kernels"]))
