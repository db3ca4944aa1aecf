"""Diagnostic: float64 Pyramid measurement on the device against the oracle at growing sizes (where does the 1e-8 come from?)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ao_oracle as O
from oracle.make_goldens import c4_test_opd
from rlao_amd import _lib as L
from rlao_amd.env import BatchedAOEnv

for D, ns, r0, L0 in ((1.6, 4, 0.13, 30.0), (1.52, 20, 0.25, 10.0), (8.0, 40, 0.13, 30.0)):
    R = 6 * ns
    geo = dict(diameter=D, nSubaperture=ns, nPixelPerSubap=6, r0=r0, L0=L0, windSpeed=[10.0], windDirection=[72.0], fractionalR0=[1.0],
               altitude=[0.0], nModes=8, modulation=0.0, nLoop=8)
    env = BatchedAOEnv(n_envs=1, device=0, dtype="f64")
    env.set_params(geo, camera="ideal", wfs_type="pyramid")
    orc = O.OracleEnv(resolution=R, diameter=D, n_subap=ns, r0=r0, L0=L0, n_modes=8, wfs_type="pyramid", m2c=env.M2C_CL, modal_cm=env.modal_CM)
    opd = c4_test_opd(R) * 0.2
    env._shard.set_atm_opd(opd.reshape(1, -1))
    env._shard.set_coefs(None)
    env.measure()
    sig = env._shard.download(L.B_SIGNAL, (1, env.nSignal))[0]
    frame = env._shard.download(L.B_FRAME, (1, env.cam_res, env.cam_res))[0]
    osig = orc.wfs.measure(opd * orc.pupil * 2 * np.pi / orc.wavelength)
    ofr = orc.wfs.frame
    print(f"R={R} nRes={env._pyr_tables.nRes}: frame max {ofr.max():.3e} |dframe|/max {np.abs(frame - ofr).max() / ofr.max():.2e}  "
          f"|dsig| {np.abs(sig - osig).max():.2e} (|sig| {np.abs(osig).max():.2e})  ref diff {np.abs(env.reference_centroids - np.concatenate([orc.wfs.referenceSignal_2D[:ns][orc.wfs.validI4Q], orc.wfs.referenceSignal_2D[ns:][orc.wfs.validI4Q]])).max():.2e}", flush=True)
    stroke = orc.wavelength / 16
    for a in (0, env.nValidAct // 2):
        col = orc.poke_signal(a, stroke)
        print(f"   imat col {a}: |d|/max {np.abs(env.imat[:, a] - col).max() / np.abs(col).max():.2e}", flush=True)
    env.close()
