"""Per-kernel device times at one or more batch sizes (GPU box), gate-path kernels first."""
import json, os, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for b in (sys.argv[1:] or ["256"]):
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--no-cpu-baseline", "--no-extras", "--batch", b,
                          "--steps", "30", "--warmup", "5"], capture_output=True, text=True)
    try:
        d = json.loads(out.stdout.strip().splitlines()[-1])
    except Exception:
        print(out.stdout[-2000:], out.stderr[-2000:]); continue
    print(f"B={b}: {d['value']:.0f} img/s ({d['serial']['value']:.0f} serial), kernel sum {d['kernel_ms_sum']*1e3:.1f} us: " +
          "  ".join(f"{k['kernel'].split(' ')[0]}={k['ms']*1e3:.1f}" for k in d["roofline_kernels"]))
