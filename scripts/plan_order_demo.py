import sys, time, os, torch
sys.path.insert(0,'.')
from mpi_amd import binding, gen
dev=torch.device('cuda')
off,adj=gen.rmat(18,16,seed=1,device=dev); n=1<<18
o,a=off.cpu().numpy(),adj.cpu().numpy()
for pat in ("path4","path3","star4"):
    for order in ("reference","leaves_last"):
        if order=="reference": os.environ["GPM_PLAN_ORDER"]="reference"
        else: os.environ.pop("GPM_PLAN_ORDER",None)
        m=binding.Matcher(1); m.upload(n,o,a); m.set_pattern(gen.PATTERNS[pat])
        t=time.time(); c=m.findMatches(count_only=True); print(pat,order,c,round(time.time()-t,3),flush=True); m.close()
