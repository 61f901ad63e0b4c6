"""The four default-fit candidates of config 3 on which /root/reference reports "Lambda correction failed" in all of its 33 protocol runs
(2^-48 input perturbations, one ulp in expm) while the HIP path returns a value: the reference ITSELF on inputs perturbed by 2^-48, 2^-40,
2^-36 and 2^-32 (16 runs each; build container only)."""
import json, os, sys, warnings
NOTE = ("config 3 under the default fit, candidates 2398 / 6761 / 7005 / 7734: llh of /root/reference itself on inputs perturbed by 2^-48, 2^-40, 2^-36 and 2^-32 "
        "(tests/parity.py: perturbed, kinds 0-15); null = Lambda correction failed")
import numpy
numpy.mat = numpy.asmatrix
ROOT = '/root/repo'
sys.path.insert(0, os.path.join(ROOT, 'tests', 'golden')); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import make_fullsize as mf
import make_golden as mg
import parity
warnings.simplefilter('ignore')
spec = os.environ.get('MISTI_POLE_SPEC', 'config3:default')      # later studies: another default-fit workload (names follow the fixtures': <workload>_default_c<candidate>)
PREFIX = spec.replace(':', '_') + '_c'
mf._W[spec] = mf.workload(spec)
w, _ = mf._W[spec]
out = {}
CANDS = [int(a) for a in sys.argv[1:]] or [2398, 6761, 7005, 7734]     # later candidates: given on the command line, merged into the file
for cand in CANDS:
    times, lam, sfs, split, mis, pus, kw, params = mf.reference_args(w, cand)
    base = mg.run_reference(times, lam, sfs, split, mis, pus, kw, params)["llh"]
    rec = {"base": base}
    for e in (48, 40, 36, 32):
        parity.PERTURB = 2.0 ** -e
        vals = [mg.run_reference(*parity.perturbed(times, lam, k), sfs, split, mis, pus, kw, params)["llh"] for k in range(16)]
        rec["2^-%d" % e] = vals
        print(cand, e, sum(v is not None for v in vals), 'of 16 runs give a value', [round(v, 6) for v in vals if v is not None][:3], flush=True)
    out[cand] = rec
parity.PERTURB = 2.0 ** -48
_path = os.path.join(ROOT, 'tests', 'golden', 'golden_pole_crossing.json')
_have = json.load(open(_path))['cases'] if os.path.exists(_path) else {}
_have.update({PREFIX + '%d' % k: v for k, v in out.items()})
json.dump({'generator': 'tests/golden/pole_reference_runs.py', 'scipy': '1.15.3', 'numpy': '2.2.6', 'note': NOTE, 'cases': _have},
          open(os.path.join(ROOT, 'tests', 'golden', 'golden_pole_crossing.json'), 'w'), indent=1)
