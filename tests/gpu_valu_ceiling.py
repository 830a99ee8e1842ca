"""VALU issue ceiling of an MI355X SIMD, measured on the box: builds and runs
profiles/valu_probe.hip and writes its JSON to gpurun_out/<tag>_valu_ceiling.json.  Not a test.

    python tests/gpu_valu_ceiling.py [tag]
"""
import json
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
tag = sys.argv[1] if len(sys.argv) > 1 else "r03"
exe, src = ROOT / "profiles" / "valu_probe", ROOT / "profiles" / "valu_probe.hip"
if not exe.exists() or exe.stat().st_mtime < src.stat().st_mtime:
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-o", str(exe), str(src)])
text = subprocess.run([str(exe)], check=True, capture_output=True, text=True, timeout=300).stdout
data = json.loads(text)
ghz = data["clock_khz"] / 1e6
for name, levels in data.items():
    if isinstance(levels, dict):
        for w, v in levels.items():
            v["cycles_per_instruction_per_simd_at_reported_clock"] = round(v["ns_per_instruction_per_simd"] * ghz, 3)
out = ROOT / "gpurun_out" / f"{tag}_valu_ceiling.json"
out.parent.mkdir(exist_ok=True)
out.write_text(json.dumps(data, indent=1))
print(json.dumps({k: {w: v["cycles_per_instruction_per_simd_at_reported_clock"] for w, v in lv.items()}
                  for k, lv in data.items() if isinstance(lv, dict)}))
