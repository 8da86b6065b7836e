import torch, time, sys
t0=time.time()
for i in range(12):
    f,t=torch.cuda.mem_get_info()
    print("%.1fs free %.1f GB of %.1f" % (time.time()-t0, f/2**30, t/2**30), flush=True)
    time.sleep(1.0)
