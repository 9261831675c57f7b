from usher_amd import synth
from usher_amd.placement import FlatTreeView
import time
st=synth.SynthTree(10_000_000, n_sites=25000, seed=1)
t0=time.time(); FlatTreeView(st.arrays); print('view', time.time()-t0)
