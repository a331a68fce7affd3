import sys
sys.path.insert(0, '.')
import eccoxide_amd as E
from tests import oracle_lib
from tests import test_ct_kernels as T
from tests import ct_model as M
from eccoxide_amd import workload as W
ora = oracle_lib.load()
with E.Engine(0) as eng:
    for curve in ('p256r1','ed25519'):
        fb, sb = T._SIZES[curve]
        vals=T._edge_scalars(curve)
        ks=T._pack(vals, sb)
        want=ora.base(curve, ks, threads=16)
        got=eng.scalarmul_base(curve, ks, ct_scan=True)
        bad=[i for i in range(len(vals)) if got[0][i*2*fb:(i+1)*2*fb]!=want[0][i*2*fb:(i+1)*2*fb] or got[1][i]!=want[1][i]]
        print(curve, len(vals), 'bad', len(bad))
        for i in bad[:12]:
            print('  ', i, hex(vals[i]), got[1][i], want[1][i], M.booth_digits(vals[i], sb, 6)[-6:], M.fixed_base_events(vals[i], sb, 6, W.order(curve)))
