import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29655", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group(backend="nccl", device_id=dev)
from gflow_amd import fit_video as FV, synthetic as S
frames = FV.upload_clip(S.make_clip(2, 96, 128, seed=0), dev)
m = FV.fit_clip(frames, dev, dict(num_points=1500, iterations_first=40, iterations_after=20, iterations_camera=10), seed=0, snapshot_interval=10)
out = FV.reduce_metrics(m, 1.0, dist, dev)
torch.cuda.synchronize(); dist.barrier()
print("nccl 1-rank ok", out["frames"], out["psnr_sum"])
dist.destroy_process_group()
