"""Golden cases (default: the default-fit campaign candidates outside at factor 3) through the HIP path with the solver trace on, against the
reference trace of the same case, solve by solve: (nfev, status), the noise bit, and the relative difference of the corrected rates (GPU box).

    python tools/trace_goldens.py [camp_m229_c5 ...]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from conftest import load_golden
from solver_trace_util import load_traces, hip_trace, KIND_OF_SITE
names = sys.argv[1:] or ['camp_m229_c5', 'camp_m353_c0', 'camp_s4_m303_c2', 'camp_s3_m417_c5', 'camp_m206_c21']
import glob, gzip, json
GOLDEN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
cases, traces = {}, load_traces()
for f in sorted(glob.glob(os.path.join(GOLDEN, 'golden_*.json'))):          # any fixture that holds the named cases (and its traces, if it has some)
    try:
        cs = load_golden(os.path.basename(f)[:-5])
    except Exception:
        continue
    if isinstance(cs, list) and cs and isinstance(cs[0], dict) and 'name' in cs[0] and 'in' in cs[0]:
        cases.update({c['name']: c for c in cs})
        tp = f[:-5] + '_traces.json.gz'
        if os.path.exists(tp):
            traces.update({c['name']: c for c in json.load(gzip.open(tp, 'rt'))['cases']})
for n in names:
    c = cases[n]; tr_ref = traces.get(n)
    llh, m, tr = hip_trace(c)
    o = c['out']
    print('==', n, 'hip', llh, 'ref', o['llh'], 'rel %.3g' % (abs(llh - o['llh']) / abs(o['llh'])), 'spread %.3g internal %.3g' % (o.get('spread') or 0, o.get('internal_spread') or 0), 'kw', c['in']['kw'], 'mi', c['in']['mi'], 'pu', c['in']['pu'], 'split', c['in']['split'])
    if not tr_ref:
        print('   no trace'); continue
    lc_ref = np.array(o['lc']); lc_hip = np.array(m.lc)
    for sv in tr_ref['solves']:
        t = sv['t']
        hip = (int(tr['nfev'][0, t]), int(tr['status'][0, t])); noise = int(tr['noise'][0, t])
        ref = (sv['nfev'], sv['status'])
        d = np.abs(lc_hip[t] - lc_ref[t]) / np.abs(lc_ref[t]) if t < len(lc_ref) else None
        flag = '' if hip == ref else '  <<<'
        print('   t %2d %-12s ref %s hip %s noise %d  lc rel diff %s%s' % (t, sv['site'], ref, hip, noise, None if d is None else '%.2e %.2e' % tuple(d), flag))
