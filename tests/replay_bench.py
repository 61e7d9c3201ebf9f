import sys, time, ctypes as C; sys.path.insert(0,'.')
import numpy as np
from oracle import binding as ob
from dump1090_rs_amd import synth, _lib
from dump1090_rs_amd.context import TRIAL_DTYPE, replay_records
from dump1090_rs_amd._lib import AdsbMsg
L=_lib.lib()
n=32*131072
iq=synth.make_iq(n, n_bursts=320*4, seed=9, n_icao=200)
Lo=ob.lib()
parts=[]
for c,off in enumerate(range(0,n,131072)):
    mb=ob.OrcMagBuf(); part=np.ascontiguousarray(iq[off:off+131072]); Lo.orc_to_mag(part.ctypes.data,len(part),C.byref(mb))
    buf=np.zeros(5*131072//8,dtype=TRIAL_DTYPE); k=Lo.orc_all_trials(C.byref(mb),c,buf.ctypes.data,len(buf)); parts.append(buf[:k].copy())
alltr=np.concatenate(parts)
msgs=replay_records(alltr)
keys=set((m.chunk,m.j) for m in msgs)
sel=np.array([ (int(r['chunk']),int(r['j_tp'])&0xFFFFFF) in keys for r in alltr])
rec=alltr[sel]
reps=int(17000/len(rec))+1
big=np.concatenate([rec]*reps)
for i in range(reps): big['chunk'][i*len(rec):(i+1)*len(rec)] += 32*i
rng=np.random.default_rng(1); big=big[rng.permutation(len(big))] if "--shuffle" in sys.argv else big[np.lexsort((big["j_tp"]>>24, big["j_tp"]&0xFFFFFF, big["chunk"]))]
out=(AdsbMsg*(1<<20))(); nout=C.c_size_t()
for _ in range(3):
    table=np.zeros(4096,np.uint32); r=big.copy()
    t=time.perf_counter(); L.adsb_replay_records(table.ctypes.data, r.ctypes.data, len(r), out, 1<<20, C.byref(nout)); dt=time.perf_counter()-t
    print(len(big), nout.value, round(dt*1e6,1),'us')
