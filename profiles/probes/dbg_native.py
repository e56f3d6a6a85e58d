import torch, sys
sys.path.insert(0, '/root/repo')
from tests import synth
from tests.test_gpu_native import _inputs, _pair
gpu = torch.device('cuda:0')
n_pool, n_steps = 6, 25
cam, frames, _ = synth.make_sequence(n_pool, 240, 320, C=64, iters=3, seed=11, closed_loop=True)
ins = _inputs(frames, gpu, static=True)
py, nat = _pair(cam, {}, gpu)
outs = []
for hp in (py, nat):
    hp.keep_extras = False
    hp.initialize(ins[0])
    sink = torch.zeros(n_steps, 7, device=gpu)
    torch.manual_seed(5)
    kps = []
    for r in hp.run((ins[(1 + k) % n_pool] for k in range(n_steps)), pose_sink=sink):
        hp.sync_pose()
        kps.append(r.kp0_uv.clone())
    torch.cuda.synchronize()
    outs.append((sink.clone(), kps))
print(torch.equal(outs[0][0], outs[1][0]))
for i,(a,b) in enumerate(zip(outs[0][1], outs[1][1])):
    print(i, a.shape, b.shape, torch.equal(a,b), (a!=b).sum().item() if a.shape==b.shape else None)
