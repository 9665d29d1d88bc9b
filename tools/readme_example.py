import re, sys
sys.path.insert(0, "/root/repo")
src = open("/root/repo/README.md").read()
code = re.search(r"```python\n(.*?)```", src, re.S).group(1)
code = code.replace("range(100)", "range(45)").replace("N, B = 1 << 20, 128", "N, B = 1 << 14, 128")
exec(code)
import torch; torch.cuda.synchronize(); print("readme example ok", float(mix.abs().max()))
