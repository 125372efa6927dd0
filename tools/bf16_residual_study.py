"""CPU study (build container, oracle only -- test infrastructure): what would an fp32 residual stream / fp32 LayerNorm
input buy the bf16 build?  

The oracle forward is re-run with bf16 ROUNDING inserted at chosen storage points and compared with the fp32 oracle:

  operands  every GEMM operand (activations entering a Linear, weights, q/k/v, softmax probabilities) rounded to bf16,
            accumulation and everything else in fp32                       -> the floor of ANY bf16-MFMA design
  +stream   additionally the residual stream / LayerNorm inputs stored in bf16 (what libmemehip.so does: GEMM
            epilogues write x + f(x) as bf16, LayerNorm reads it)          -> the shipped bf16 build
  fp16 rows the same two modes with IEEE half (11-bit significand)

    python tools/bf16_residual_study.py [--full]      (--full: config 3 at batch 4; default: the 2-layer tiny model)
"""
import math
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from oracle import meme_oracle as O


def make_q(dtype):
    return (lambda x: x.to(dtype).float()) if dtype is not None else (lambda x: x)


def mha(x, p, L, names, heads, add_mask, q):
    B, S, D = x.shape
    dh = D // heads
    xq = q(x)
    def lin(n):
        return F.linear(xq, q(p[L + names[n] + ".weight"]), p[L + names[n] + ".bias"])
    Q, K, V = (q(lin(n)).view(B, S, heads, dh).transpose(1, 2) for n in ("q", "k", "v"))
    s = torch.matmul(Q, K.transpose(-1, -2)) / math.sqrt(dh)
    if add_mask is not None:
        s = s + add_mask
    P = q(torch.softmax(s, dim=-1))
    return q(torch.matmul(P, V).transpose(1, 2).reshape(B, S, D))


def text_tower(p, ids, mask, c, q, qs):
    pfx = "bert."
    S = ids.shape[1]
    x = F.embedding(ids, p[pfx + "embeddings.word_embeddings.weight"]) + p[pfx + "embeddings.position_embeddings.weight"][:S][None]
    if c.type_vocab > 0:
        x = x + p[pfx + "embeddings.token_type_embeddings.weight"][0][None, None]
    x = qs(F.layer_norm(qs(x), (c.hidden,), p[pfx + "embeddings.LayerNorm.weight"], p[pfx + "embeddings.LayerNorm.bias"], c.ln_eps))
    add_mask = (1.0 - mask.float())[:, None, None, :] * torch.finfo(torch.float32).min
    names = {"q": "attention.self.query", "k": "attention.self.key", "v": "attention.self.value"}
    for i in range(c.layers):
        L = f"{pfx}encoder.layer.{i}."
        ctx = mha(x, p, L, names, c.heads, add_mask, q)
        a = F.linear(ctx, q(p[L + "attention.output.dense.weight"]), p[L + "attention.output.dense.bias"])
        x = qs(F.layer_norm(qs(x + a), (c.hidden,), p[L + "attention.output.LayerNorm.weight"], p[L + "attention.output.LayerNorm.bias"], c.ln_eps))
        h = q(F.gelu(F.linear(q(x), q(p[L + "intermediate.dense.weight"]), p[L + "intermediate.dense.bias"])))
        o = F.linear(h, q(p[L + "output.dense.weight"]), p[L + "output.dense.bias"])
        x = qs(F.layer_norm(qs(x + o), (c.hidden,), p[L + "output.LayerNorm.weight"], p[L + "output.LayerNorm.bias"], c.ln_eps))
    return x


def image_tower(p, image, c, q, qs):
    pfx = "image_model."
    B, D = image.shape[0], c.hidden
    w = p[pfx + "embeddings.patch_embeddings.projection.weight"].reshape(D, -1)
    x = F.linear(q(O.patchify(image, c.patch)), q(w), p[pfx + "embeddings.patch_embeddings.projection.bias"])
    x = torch.cat([p[pfx + "embeddings.cls_token"].expand(B, -1, -1), x], dim=1) + p[pfx + "embeddings.position_embeddings"]
    x = qs(x)
    names = {"q": "attention.attention.query", "k": "attention.attention.key", "v": "attention.attention.value"}
    for i in range(c.layers):
        L = f"{pfx}encoder.layer.{i}."
        h = F.layer_norm(x, (D,), p[L + "layernorm_before.weight"], p[L + "layernorm_before.bias"], c.ln_eps)
        ctx = mha(h, p, L, names, c.heads, None, q)
        x = qs(x + F.linear(ctx, q(p[L + "attention.output.dense.weight"]), p[L + "attention.output.dense.bias"]))
        h = q(F.layer_norm(x, (D,), p[L + "layernorm_after.weight"], p[L + "layernorm_after.bias"], c.ln_eps))
        h = q(F.gelu(F.linear(h, q(p[L + "intermediate.dense.weight"]), p[L + "intermediate.dense.bias"])))
        x = qs(x + F.linear(h, q(p[L + "output.dense.weight"]), p[L + "output.dense.bias"]))
    return F.layer_norm(x, (D,), p[pfx + "layernorm.weight"], p[pfx + "layernorm.bias"], c.ln_eps)


def forward(p, text, image, mask, cfg, q, qs):
    t = text_tower(p, text, mask, cfg.text, q, qs)[:, 0]
    t = F.linear(t, p["bert_fc.weight"], p["bert_fc.bias"])
    v = F.linear(image_tower(p, image, cfg.image, q, qs)[:, 0], p["image_fc.weight"], p["image_fc.bias"])
    return F.linear(F.linear(torch.cat((t, v), 1), p["fusion_fc.weight"], p["fusion_fc.bias"]), p["output_fc.weight"], p["output_fc.bias"])


def main():
    torch.set_num_threads(8)
    full = "--full" in sys.argv
    cfg = O.config3("cls") if full else O.tiny_config("cls")
    B, S = (4, 128) if full else (4, 16)
    ident = make_q(None)
    rows = []
    for seed in ((3, 4) if full else (1, 2, 3, 4, 5, 6)):
        p = O.init_params(cfg, seed)
        text, image, mask, _ = O.synthetic_batch(cfg, B, S, seed=100 + seed)
        with torch.no_grad():
            ref = forward(p, text, image, mask, cfg, ident, ident)
            assert float((ref - O.forward(p, text, image, mask, cfg)).abs().max()) < 1e-5
            for name, dt in (("bf16", torch.bfloat16), ("fp16", torch.float16)):
                q = make_q(dt)
                e_op = float((forward(p, text, image, mask, cfg, q, ident) - ref).abs().max())
                e_all = float((forward(p, text, image, mask, cfg, q, q) - ref).abs().max())
                rows.append((seed, name, e_op, e_all))
                print(f"seed {seed} {name}: operands-only (fp32 residual stream) {e_op:.2e}   + 16-bit stream (shipped design) {e_all:.2e}", flush=True)
    for name in ("bf16", "fp16"):
        ops = [r[2] for r in rows if r[1] == name]
        alls = [r[3] for r in rows if r[1] == name]
        print(f"{name}: operands-only max {max(ops):.2e} mean {sum(ops) / len(ops):.2e};  with 16-bit stream max {max(alls):.2e} mean {sum(alls) / len(alls):.2e}")


if __name__ == "__main__":
    main()
