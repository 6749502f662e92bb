import sys, numpy as np, glob, os
d = sys.argv[1]
files = sorted(glob.glob(os.path.join(d, 'dump_*.npz')))
data = [np.load(f) for f in files]
base = data[0]
for i in range(1, len(data)):
    bad = []
    for k in base.files:
        a, b = base[k].astype(np.float64), data[i][k].astype(np.float64)
        e = np.linalg.norm(a - b) / (np.linalg.norm(a) + 1e-30)
        if e > 3e-6: bad.append('%s %.1e' % (k, e))
    print(os.path.basename(files[i]), 'vs', os.path.basename(files[0]), ':', bad)
import json
offs = json.load(open(os.path.join(d, 'offsets.json')))
def name_of(net, idx):
    best = None
    for k, o in offs[net].items():
        if o <= idx and (best is None or o > offs[net][best]): best = k
    return '%s[%d]' % (best, idx - offs[net][best])
gxs = [float(np.abs(x['gx']).sum()) for x in data]
print('gx sums', gxs)
for i in range(1, len(data)):
    for key, net in (('post.DIp', 'DI'), ('post.DVp', 'DV'), ('pre.DIp', 'DI'), ('pre.DVp', 'DV'), ('DIg', 'DI'), ('DVg', 'DV')):
        a, b = base[key], data[i][key]
        dd = np.abs(a - b)
        idx = np.argsort(-dd)[:6]
        print(os.path.basename(files[i]), key, [(name_of(net, int(j)), '%.2e' % dd[j], '%.3e' % a[j], '%.3e' % b[j]) for j in idx if dd[j] > 1e-6])
    if i >= 3: break
