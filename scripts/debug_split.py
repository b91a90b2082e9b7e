import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from flexynesis_amd import ops
dev=torch.device("cuda:0")
x=torch.arange(2*40,dtype=torch.float32,device=dev).reshape(2,40)
hi,lo=ops.new_split(2,40,dev)
ops.split_bf16(ops.IMMEDIATE,hi,lo,x)
torch.cuda.synchronize()
print(x[0,:16]); print(hi[0,:16].float()); print(lo[0,:16].float()); print(hi[1,:16].float()); print(hi[0,32:48].float())
