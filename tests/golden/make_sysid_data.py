"""Writes tests/golden/sysid_data.npz: the system-identification series that BASELINE configs 1 and 2 are quoted on (actuator.mat:
variables 'u', 'p'; ballbeam.dat: columns 0, 1) and the two others whose notebooks print evaluation rates (identificationExample5.mat,
drive.mat), taken from the reference's data files under datasets/system_identification/ with the train/test split points of
autoreg/benchmark/tasks.py:125-195.  Run once in the build container (the reference tree
does not travel to the GPU box); only numbers are stored.

    python tests/golden/make_sysid_data.py [/root/reference]
"""
import os
import sys

import numpy as np
import scipy.io


def main():
    ref = sys.argv[1] if len(sys.argv) > 1 else '/root/reference'
    d = os.path.join(ref, 'datasets', 'system_identification')
    act = scipy.io.loadmat(os.path.join(d, 'actuator.mat'))
    bb = np.loadtxt(os.path.join(d, 'ballbeam.dat'))
    ie5 = scipy.io.loadmat(os.path.join(d, 'identificationExample5.mat'))     # tasks.py:125-138: 'u', 'y' train; 'uNew', 'yNew' test
    drv = scipy.io.loadmat(os.path.join(d, 'drive.mat'))                      # tasks.py:141-158 (Drive): 'u1', 'z1', split 250
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sysid_data.npz')
    np.savez_compressed(out,
                        actuator_u=np.asarray(act['u'], dtype=np.float64).reshape(-1, 1),
                        actuator_y=np.asarray(act['p'], dtype=np.float64).reshape(-1, 1),
                        actuator_split=np.int64(512),
                        ballbeam_u=np.asarray(bb[:, 0], dtype=np.float64).reshape(-1, 1),
                        ballbeam_y=np.asarray(bb[:, 1], dtype=np.float64).reshape(-1, 1),
                        ballbeam_split=np.int64(500),
                        ie5_u=np.concatenate([np.asarray(ie5['u'], dtype=np.float64).reshape(-1, 1),
                                              np.asarray(ie5['uNew'], dtype=np.float64).reshape(-1, 1)]),
                        ie5_y=np.concatenate([np.asarray(ie5['y'], dtype=np.float64).reshape(-1, 1),
                                              np.asarray(ie5['yNew'], dtype=np.float64).reshape(-1, 1)]),
                        ie5_split=np.int64(np.asarray(ie5['u']).size),
                        drive_u=np.asarray(drv['u1'], dtype=np.float64).reshape(-1, 1),
                        drive_y=np.asarray(drv['z1'], dtype=np.float64).reshape(-1, 1),
                        drive_split=np.int64(250))
    z = np.load(out)
    print({k: z[k].shape for k in z.files})


if __name__ == '__main__':
    main()
