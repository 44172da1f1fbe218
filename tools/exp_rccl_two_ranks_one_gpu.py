"""Does RCCL accept two ranks on ONE device?  (Run on a 1-GPU box: python tools/exp_rccl_two_ranks_one_gpu.py)
Spawns two processes that build the library's communicator (igmc_comm_create) on cuda:0; prints what happens."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WORKER = r'''
import os, sys
sys.path.insert(0, %r)
import torch, torch.distributed as dist
rank = int(os.environ['RANK'])
torch.cuda.set_device(0)
dist.init_process_group(backend='gloo', init_method='tcp://127.0.0.1:29651', rank=rank, world_size=2)
from igmc_amd import _lib, parallel
try:
    c = parallel.GradComm(_lib.load(), 0)
    print('rank', rank, 'communicator created:', c.info())
    t = torch.ones(8, device='cuda')
    c.all_reduce_(t, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    print('rank', rank, 'all-reduce ->', t[:2].tolist())
except Exception as e:
    print('rank', rank, 'FAILED:', str(e)[:300])
''' % ROOT

if __name__ == '__main__':
    ps = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE='2', HSA_ENABLE_IPC_MODE_LEGACY='0', NCCL_DEBUG='WARN')
        ps.append(subprocess.Popen([sys.executable, '-c', WORKER], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    for p in ps:
        try:
            print(p.communicate(timeout=120)[0].decode()[-1500:])
        except subprocess.TimeoutExpired:
            p.kill()
            print('TIMEOUT')
