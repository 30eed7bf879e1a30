"""Long-run soak: 3000 generations at cfg2 and cfg3 twice each; parents, accessory matrix and distances must replay
bit for bit (deterministic replay under --seed) and no device queue may overflow.  Prints generations/s."""
import os
import sys
import time
import zlib

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import pansim_amd as pa
crc = lambda a: zlib.crc32(np.ascontiguousarray(a).view(np.uint8))
for kw in (dict(), dict(HR_rate=0.5, HGT_rate=0.5)):
    out = []
    for rep in range(2):
        sim = pa.Simulation(pa.make_params(seed=7, n_gen=3000, max_distances=1000, **kw))
        t = time.time(); sim.run(3000); sim.sync(); dt = time.time() - t
        core_d, acc_d = sim.final_distances()
        out.append((crc(sim.last_parents()), crc(sim.pan_genome.read_matrix()), crc(core_d), crc(acc_d)))
        sim.close()
    print(kw, out[0] == out[1], out[0], round(3000 / dt, 1), "gen/s")
