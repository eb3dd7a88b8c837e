import os,sys,time,torch
sys.path.insert(0,os.getcwd())
import kateth_amd
n=4096
s=kateth_amd.Setup.load_json("tests/golden/trusted_setup_4096.json", window_bits=12)
d_blobs=torch.empty(n*131072,dtype=torch.uint8,device="cuda"); d_c=torch.empty(n*48,dtype=torch.uint8,device="cuda"); d_st=torch.empty(n,dtype=torch.int32,device="cuda")
s.synth_blobs_dev(0x4844,0,n,d_blobs.data_ptr())
for chunk in (4096,2048,1024,512,256):
    def run():
        for b in range(0,n,chunk):
            s.blob_to_commitment_batch_dev(d_blobs.data_ptr()+b*131072,chunk,d_c.data_ptr()+b*48,d_st.data_ptr()+b)
        torch.cuda.synchronize()
    run(); t0=time.perf_counter(); run(); run(); dt=(time.perf_counter()-t0)/2
    print("chunk %d: %.1f ms"%(chunk,dt*1e3),flush=True)
