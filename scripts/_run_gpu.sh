mkdir -p gpurun_out/r2h; O=gpurun_out/r2h
python -m pytest tests/test_gpu_fused.py -m gpu -x -q -k rejects 2>&1 | tail -60 > $O/pytest.txt
grep -n "Error\|raise\|assert\|^E " $O/pytest.txt | head -30
python - <<'PY'
import sys, torch, math
sys.path.insert(0,'.')
import bench
from gaussian_splatting_amd import fused, _hip
from gaussian_splatting_amd.synthetic import DEFAULTS, WORKLOADS, make_scene
dev=torch.device('cuda',0)
N,W,H,deg=WORKLOADS['D']
g,cam,T=make_scene(N,W,H,deg,seed=0,device=dev)
d=DEFAULTS
def sv(T):
    f=fused.preprocess_forward(g.xyz,g.quaternion,g.scale,g.opacity,g.rgb,g.sh,T,cam.K,W,H,d['near_thresh'],d['far_thresh'],d['cull_mask_padding'],d['mh_dist'],None,0)
    return f.S,f.V
print('identity',sv(torch.eye(4,device=dev)))
for tz in (-3,-2,-1,1,2,3,5):
    M=torch.eye(4,device=dev); M[2,3]=tz; print('tz',tz,sv(M))
for tx in (1,2,4):
    M=torch.eye(4,device=dev); M[0,3]=tx; print('tx',tx,sv(M))
for yaw in (4,8,15):
    a=math.radians(yaw); M=torch.eye(4,device=dev); M[0,0]=math.cos(a); M[0,2]=math.sin(a); M[2,0]=-math.sin(a); M[2,2]=math.cos(a); print('yaw',yaw,sv(M))
PY
