import time, logging, os, sys, torch
sys.path.insert(0, os.getcwd())
from srgd_amd.config import load_config
from srgd_amd.model import get_model
from srgd_amd.synth import synth_state_dict
conf = load_config("conf/conditional_continuous_linear_df8kost_dim128.yaml")
t0=time.time(); s = get_model(conf, logging.getLogger("x")).module; t1=time.time()
schema = {k: tuple(v.shape) for k, v in s.state_dict().items()}
sd = synth_state_dict(schema, seed=0); t2=time.time()
s.load_state_dict(sd, strict=True); s=s.eval().to("cuda"); torch.cuda.synchronize(); t3=time.time()
for prec in ("bf16","fp32"):
    ta=time.time(); e = s.model.engine(prec); torch.cuda.synchronize(); tb=time.time()
    print(prec, "engine create+pack+upload s:", round(tb-ta,2))
print("get_model", round(t1-t0,2), "synth", round(t2-t1,2), "load_state_dict+to(cuda)", round(t3-t2,2))
