"""Where the organizers' model (DistilBERT + ResNet-50) deviates from the fp32 CPU composition: per tower, batch 1 of the
reference-run fixture; the ResNet side also against the oracle with the 16-bit STORAGE rounding inserted (fp32 arithmetic)."""
import os, sys, tempfile
import numpy as np, torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import multimodal_propaganda_meme_classification_amd as pkg
from oracle import ref_env as E, resnet_oracle as R, meme_oracle as O
from PIL import Image

z = np.load(os.path.join(E.GOLDEN, "ref_organizers_2c.npz"))
layers = tuple(int(x) for x in z["cfg_resnet_layers"]); V = int(z["vocab_size"]); B = 8
state = E.organizers_state(V, 6, layers, int(z["cfg_seed"]))
work = tempfile.mkdtemp(); E.write_dataset(work)
tf = E.organizers_transform()
image = torch.stack([tf(Image.open(os.path.join(work, r["img_path"])).convert("RGB")) for r in E.records24()[:B]])
text, mask = torch.from_numpy(z["ds_text"][:B]), torch.from_numpy(z["ds_text_mask"][:B])
p_res = {k[7:]: v for k, v in state.items() if k.startswith("resnet.")}
torch.set_num_threads(16)
with torch.no_grad():
    ref32 = R.resnet_forward(p_res, R.new_bn_state(p_res), image, layers, training=True)
    ref16 = R.resnet_forward(p_res, R.new_bn_state(p_res), image, layers, training=True, storage=torch.float16)
print("resnet logits: |ref32| max", float(ref32.abs().max()), "std", float(ref32.std()), "| storage-rounded oracle vs fp32:", float((ref16 - ref32).abs().max()))
for dt in ("fp16", "bf16"):
    net = pkg.ResNet50(num_classes=1000, compute_dtype=dt, layers=layers)
    sd = net.state_dict()
    for k, v in p_res.items():
        sd[k].copy_(v)
    net.cuda().train()
    with torch.no_grad():
        got = net(image.cuda()).float().cpu()
    print(f"[{dt}] hip resnet vs fp32 oracle: {float((got - ref32).abs().max()):.3e}; vs storage-rounded(fp16) oracle: {float((got - ref16).abs().max()):.3e}")
    if dt == "fp16":
        w, b = state["resnet_fc.weight"], state["resnet_fc.bias"]
        print("   after resnet_fc:", float((F.linear(got, w, b) - F.linear(ref32, w, b)).abs().max()), " storage-rounded:", float((F.linear(ref16, w, b) - F.linear(ref32, w, b)).abs().max()))
tcfg = O.TextConfig(vocab_size=V, hidden=768, layers=6, heads=12, intermediate=3072, max_position=512, type_vocab=0)
back = {E.bert_to_distil_name(k): k for k in O._text_shapes(tcfg, pfx="")}
p_text = {"bert." + back[k[5:]]: v for k, v in state.items() if k.startswith("bert.")}
with torch.no_grad():
    t_ref = O.text_tower(p_text, text, mask, tcfg)[:, -1]
tc = pkg.TextConfig(vocab_size=V, hidden=768, layers=6, heads=12, intermediate=3072, max_position=512, type_vocab=0)
enc = pkg.TextEncoder(tc, pool="last", compute_dtype="fp16", naming="distilbert")
enc.load_state_dict({k[5:]: v for k, v in state.items() if k.startswith("bert.")})
enc.cuda()
with torch.no_grad():
    t_got = enc(text.cuda(), mask.cuda()).float().cpu()
print("text tower (last position) vs oracle:", float((t_got - t_ref).abs().max()), "|ref| max", float(t_ref.abs().max()))
