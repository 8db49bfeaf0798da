import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import bench as B
import ptranking_amd as pa
dev="cuda:0"
torch.manual_seed(1)
r = pa.LambdaRank(sf_para_dict=B.sf_para_dict(136), model_para_dict={"sigma":1.0}, gpu=True, device=dev); r.init(); r.train_mode()
gen = torch.Generator(device=dev).manual_seed(1)
X,Y = B.synth_batch(gen, 4096, 128, 136, dev)
t_start=time.perf_counter()
for blk in range(12):
    t0=time.perf_counter()
    for i in range(50):
        r.train_op(X,Y,epoch_k=1,presort=True,label_type=pa.LABEL_TYPE.MultiLabel)
    t1=time.perf_counter()
    torch.cuda.synchronize()
    t2=time.perf_counter()
    print(f"t={t0-t_start:6.2f}s  enqueue {1e3*(t1-t0)/50:.3f} ms/step   wall {1e3*(t2-t0)/50:.3f} ms/step", flush=True)
