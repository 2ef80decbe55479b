import re, subprocess, threading, time, torch, json
samples=[]; stop=False
def poll():
    while not stop:
        out = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
        sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", out); pw = re.search(r"Power \(W\): ([\d.]+)", out)
        if sclk and pw: samples.append((time.perf_counter(), int(sclk.group(1)), float(pw.group(1))))
threading.Thread(target=poll, daemon=True).start()
a = torch.empty(1<<30, dtype=torch.uint8, device="cuda"); b = torch.empty_like(a)
x = torch.randn(1<<28, device="cuda"); 
def phase(name, fn, bytes_per):
    fn(); torch.cuda.synchronize(); t0=time.perf_counter(); it=0
    while time.perf_counter()-t0 < 3.0:
        fn(); it+=1
        if it%8==0: torch.cuda.synchronize()
    torch.cuda.synchronize(); t1=time.perf_counter()
    mid=[(s,p) for (t,s,p) in samples if t0+1.0<t<t1]
    print(name, json.dumps({"GBps": it*bytes_per/(t1-t0)/1e9, "sclk": sum(s for s,_ in mid)/max(1,len(mid)), "W": sum(p for _,p in mid)/max(1,len(mid))}), flush=True)
    time.sleep(1.0)
phase("copy 1GiB (read+write)", lambda: b.copy_(a), 2*(1<<30))
phase("read-only sum fp32", lambda: x.sum(), 4*(1<<28))
phase("fill (write-only)", lambda: b.fill_(1), (1<<30))
stop=True
