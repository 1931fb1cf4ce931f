import torch, time
x = torch.empty(int(5.37e9)//4, device="cuda")
def t(fn, n=10):
    fn(); torch.cuda.synchronize()
    a,b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n
ms = t(lambda: x.zero_())
print("fill 5.37 GB: %.3f ms = %.2f TB/s" % (ms, 5.37e9/ms/1e9))
y = torch.empty_like(x[:x.numel()//2]); 
ms = t(lambda: y.copy_(x[:y.numel()]))
print("copy 2.68 GB -> 2.68 GB: %.3f ms = %.2f TB/s (read+write)" % (ms, 5.37e9/ms/1e9))
