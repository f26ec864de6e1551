"""Writes tests/golden/sysid_data.npz: the two system-identification series that BASELINE configs 1 and 2 are quoted on, taken
from the reference's data files (datasets/system_identification/actuator.mat: variables 'u', 'p'; ballbeam.dat: columns 0, 1)
with the train/test split points of autoreg/benchmark/tasks.py:141-177.  Run once in the build container (the reference tree
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
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'sysid_data.npz')
    np.savez_compressed(out,
                        actuator_u=np.asarray(act['u'], dtype=np.float64).reshape(-1, 1),
                        actuator_y=np.asarray(act['p'], dtype=np.float64).reshape(-1, 1),
                        actuator_split=np.int64(512),
                        ballbeam_u=np.asarray(bb[:, 0], dtype=np.float64).reshape(-1, 1),
                        ballbeam_y=np.asarray(bb[:, 1], dtype=np.float64).reshape(-1, 1),
                        ballbeam_split=np.int64(500))
    z = np.load(out)
    print({k: z[k].shape for k in z.files})


if __name__ == '__main__':
    main()
