import sys, traceback, collections, torch
sys.path.insert(0, "/root/repo")
import bench
from iseg_amd import functional as F
seen = collections.Counter()
orig = torch.Tensor.contiguous
def patched(self, *a, **k):
    if not self.is_contiguous():
        st = traceback.extract_stack(limit=6)
        seen[" <- ".join(f"{f.filename.split('/')[-1]}:{f.lineno}" for f in st[:-1][::-1][:4])] += 1
    return orig(self, *a, **k)
torch.Tensor.contiguous = patched
origcopy = torch.Tensor.copy_
def pcopy(self, src, *a, **k):
    if self.is_cuda and src.is_cuda:
        st = traceback.extract_stack(limit=6)
        seen["copy_ " + " <- ".join(f"{f.filename.split('/')[-1]}:{f.lineno}" for f in st[:-1][::-1][:4])] += 1
    return origcopy(self, src, *a, **k)
torch.Tensor.copy_ = pcopy
def wrap(name, obj=torch.Tensor):
    o = getattr(obj, name)
    def f(*a, **k):
        r = o(*a, **k)
        t = a[0] if a and isinstance(a[0], torch.Tensor) else r
        if isinstance(t, torch.Tensor) and t.is_cuda:
            st = traceback.extract_stack(limit=6)
            seen[name + " " + " <- ".join(f"{f.filename.split('/')[-1]}:{f.lineno}" for f in st[:-1][::-1][:4])] += 1
        return r
    setattr(obj, name, f)
for n in ("clone", "fill_", "zero_"):
    wrap(n)
for n in ("zeros", "ones", "full", "zeros_like", "cat"):
    wrap(n, torch)
class A: pass
args = A(); args.gpus=1; args.batch=16; args.size=512; args.fp32=False
strategy, model, trainer = bench.build_trainer(args)
from iseg_amd.data import synthetic_batch
x, y = synthetic_batch(16, 512, 512, seed=100)
x, y = x.cuda(), y.cuda()
for i in range(3):
    if i == 2: seen.clear()
    trainer.train_step(x, y)
torch.cuda.synchronize()
for k, v in seen.most_common(): print(v, k)
