"""Config-3-sized model (8 x T=512, 59-D outputs, GRU encoder): 20 SVI steps, for rocprofv3 --kernel-trace."""
import sys
sys.path.insert(0, '.'); sys.path.insert(0, 'tools')
from bench_configs import data
from rgp_amd.model import DeepAutoreg_rnn
U, Y = data(8, 512, 1, 59)
m = DeepAutoreg_rnn([0, 10], Y, U=U, U_win=10, nDims=[59, 3], num_inducing=100, rnn_type='gru', rnn_hidden_dims=[20],
                    inference_method='svi', mb_inf_tot_data_size=8, seed=1)
for _ in range(20): m.svi_step(slice(0, 8), lr=1e-3)
m.handle.sync()
