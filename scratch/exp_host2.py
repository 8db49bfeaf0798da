import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench as B
import ptranking_amd as pa
from ptranking_amd import _lib
dev="cuda:0"
torch.manual_seed(1)
r = pa.LambdaRank(sf_para_dict=B.sf_para_dict(136), model_para_dict={"sigma":1.0}, gpu=True, device=dev); r.init(); r.train_mode()
gen = torch.Generator(device=dev).manual_seed(1)
batches=[B.synth_batch(gen, 4096, 128, 136, dev) for _ in range(4)]
def run(tag, nb, timing, acc):
    el = torch.zeros((), device=dev)
    for rep in range(3):
        _lib.TIMING = {} if timing else None
        torch.cuda.synchronize(); t0=time.perf_counter()
        for i in range(50):
            X,Y = batches[i % nb]
            l,_ = r.train_op(X,Y,epoch_k=1,presort=True,label_type=pa.LABEL_TYPE.MultiLabel)
            if acc: el += l.detach()
        t1=time.perf_counter(); torch.cuda.synchronize(); t2=time.perf_counter()
        _lib.TIMING=None
        print(f"{tag:28s} enqueue {1e3*(t1-t0)/50:.3f}  wall {1e3*(t2-t0)/50:.3f} ms/step", flush=True)
run("1 batch", 1, False, False)
run("4 batches", 4, False, False)
run("4 batches + acc", 4, False, True)
run("4 batches + acc + events", 4, True, True)
run("1 batch + events", 1, True, False)
