import sys, time
sys.path.insert(0, '/root/repo')
from enzymm_amd import jess_run, synthetic, template
matcher = jess_run.Matcher(template.load_packed_library())
base = synthetic.structures(1000)
mols = [base[i % len(base)].copy() for i in range(12000)]
matcher.run(mols[:2000])
t0 = time.perf_counter(); matcher.run(mols); print("rate", 12000 / (time.perf_counter() - t0), file=sys.stderr)
