"""A tiny fake huggingface cache holding one diffusers-layout snapshot (random weights of the tiny topology): what
``create_mvd_pipeline("<org>/<name>", cache_dir=...)`` must be able to resolve offline (mvd_unet.py:411-415, infer.py:33-44)."""
import json
import os

import torch

REPO = "fake-org/tiny-sd-2-1"
REV = "0123456789abcdef0123456789abcdef01234567"
VOCAB_WORDS = ["a", "photo", "of", "chair", "red", "the", "view", "front"]


def build_fake_hf_cache(root: str, with_text_encoder: bool = True, with_vae: bool = True, seed: int = 0):
    """-> (cache_dir, snapshot_dir, {component: state_dict}).  Layout: <root>/hub/models--fake-org--tiny-sd-2-1/{refs/main,
    snapshots/<REV>/{unet,vae,scheduler,text_encoder,tokenizer}}."""
    from safetensors.torch import save_file
    from mvd_amd.config import UNetConfig
    from mvd_amd.unet_params import UNet2DConditionParams
    cache = os.path.join(root, "hub")
    repo = os.path.join(cache, "models--" + REPO.replace("/", "--"))
    snap = os.path.join(repo, "snapshots", REV)
    os.makedirs(os.path.join(repo, "refs"), exist_ok=True)
    open(os.path.join(repo, "refs", "main"), "w").write(REV)
    torch.manual_seed(seed)
    cfg = UNetConfig.tiny()
    sds = {}
    # ---- unet
    os.makedirs(os.path.join(snap, "unet"))
    json.dump({"_class_name": "UNet2DConditionModel", "in_channels": cfg.in_channels, "out_channels": cfg.out_channels,
               "block_out_channels": list(cfg.block_out_channels), "layers_per_block": cfg.layers_per_block,
               "attention_head_dim": list(cfg.num_heads), "cross_attention_dim": cfg.cross_attention_dim,
               "norm_num_groups": cfg.norm_num_groups, "norm_eps": cfg.norm_eps, "sample_size": cfg.sample_size,
               "use_linear_projection": True}, open(os.path.join(snap, "unet", "config.json"), "w"))
    unet = UNet2DConditionParams(cfg)
    sds["unet"] = {k: v.detach().clone() for k, v in unet.state_dict().items()}
    save_file(sds["unet"], os.path.join(snap, "unet", "diffusion_pytorch_model.safetensors"))
    # ---- scheduler (SD-2.1's published config)
    os.makedirs(os.path.join(snap, "scheduler"))
    json.dump({"_class_name": "DDIMScheduler", "num_train_timesteps": 1000, "beta_start": 0.00085, "beta_end": 0.012,
               "beta_schedule": "scaled_linear", "prediction_type": "v_prediction", "steps_offset": 1,
               "clip_sample": False, "set_alpha_to_one": False}, open(os.path.join(snap, "scheduler", "scheduler_config.json"), "w"))
    # ---- vae
    if with_vae:
        from mvd_amd.vae import AutoencoderKLHIP, VAEConfig
        os.makedirs(os.path.join(snap, "vae"))
        vc = dict(in_channels=3, latent_channels=4, block_out_channels=[64, 128], layers_per_block=1, norm_num_groups=32,
                  scaling_factor=0.18215)
        json.dump({"_class_name": "AutoencoderKL", **vc}, open(os.path.join(snap, "vae", "config.json"), "w"))
        vae = AutoencoderKLHIP(VAEConfig(**vc))
        sds["vae"] = {k: v.detach().clone() for k, v in vae.state_dict().items()}
        save_file(sds["vae"], os.path.join(snap, "vae", "diffusion_pytorch_model.safetensors"))
    # ---- CLIP text encoder + tokenizer (tiny, random): only when transformers is importable
    if with_text_encoder:
        from transformers import CLIPTextConfig, CLIPTextModel
        tok = os.path.join(snap, "tokenizer")
        os.makedirs(tok)
        vocab = {"<|startoftext|>": 0, "<|endoftext|>": 1}
        for ch in "abcdefghijklmnopqrstuvwxyz":
            vocab[ch] = len(vocab)
            vocab[ch + "</w>"] = len(vocab)
        merges = ["#version: 0.2"]
        for w in VOCAB_WORDS:                      # merge each word's characters left to right into one token "<word></w>"
            parts = list(w[:-1]) + [w[-1] + "</w>"]
            while len(parts) > 1:
                merges.append(f"{parts[0]} {parts[1]}")
                parts = [parts[0] + parts[1]] + parts[2:]
                if parts[0] not in vocab:
                    vocab[parts[0]] = len(vocab)
        json.dump(vocab, open(os.path.join(tok, "vocab.json"), "w"))
        open(os.path.join(tok, "merges.txt"), "w").write("\n".join(dict.fromkeys(merges)) + "\n")
        json.dump({"model_max_length": 77, "tokenizer_class": "CLIPTokenizer", "bos_token": "<|startoftext|>",
                   "eos_token": "<|endoftext|>", "unk_token": "<|endoftext|>", "pad_token": "<|endoftext|>"},
                  open(os.path.join(tok, "tokenizer_config.json"), "w"))
        tc = CLIPTextConfig(vocab_size=len(vocab), hidden_size=cfg.cross_attention_dim, intermediate_size=256, num_hidden_layers=2,
                            num_attention_heads=4, max_position_embeddings=77, bos_token_id=0, eos_token_id=1, pad_token_id=1)
        te = CLIPTextModel(tc)
        te.save_pretrained(os.path.join(snap, "text_encoder"))
        sds["text_encoder"] = {k: v.detach().clone() for k, v in te.state_dict().items()}
    return cache, snap, sds
