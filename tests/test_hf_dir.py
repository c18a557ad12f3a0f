"""A real model DIRECTORY end to end, offline (reference surface: ``VLM_INSPECTOR_MODEL`` /
``model_id`` of config/models.yaml:5-18, utils/config.py:42-76): config.json / generation_config.json /
preprocessor_config.json / tokenizer.json exactly as transformers 5.15 ``save_pretrained`` wrote them for tiny seeded
qwen2_vl, qwen2_5_vl and mllama models (tests/golden/gen_hf_dir.py), model.safetensors under the tensor names it used.

CPU part: ``from_hf_dir`` (nested text_config, rope_parameters, eos ids from generation_config.json, pixel bounds and
normalisation from preprocessor_config.json), the tensor-name normalisation of both weight loaders, and the chat layout
of HFTokenizer / LlamaHFTokenizer against the ids HF's own chat template + processor produced for the same request.
GPU part (-m gpu): ``LocalVLMClient().chat.completions.create(model=<dir>, ...)`` against the logits / tokens / reply the
published transformers model gave for that request."""
import base64
import io
import json
import os

import numpy as np
import pytest
import torch
from PIL import Image

from helpers import HF_DIRS, hf_expected, materialize_hf_dir

QWEN = ["qwen2vl_tiny", "qwen25vl_tiny"]


def _messages(exp):
    uri = "data:image/png;base64," + base64.b64encode(exp["png"].tobytes()).decode()
    return [{"role": "user", "content": [{"type": "text", "text": str(exp["prompt"])},
                                         {"type": "image_url", "image_url": {"url": uri}}]}]


@pytest.mark.parametrize("family", QWEN)
def test_qwen_dir_config_tokenizer_and_chat_ids(family, tmp_path):
    from vision_inspection_system_amd.client import local_model_type
    from vision_inspection_system_amd.config import Qwen2VLConfig
    from vision_inspection_system_amd.image_processing import decode_data_uri, target_size
    from vision_inspection_system_amd.tokenizer import HFTokenizer, build_chat_ids
    d = materialize_hf_dir(family, tmp_path / family)
    exp = hf_expected(family)
    assert local_model_type(d) == ("qwen2_vl" if family == "qwen2vl_tiny" else "qwen2_5_vl")
    cfg = Qwen2VLConfig.from_hf_dir(d)
    tiny = Qwen2VLConfig.tiny() if family == "qwen2vl_tiny" else Qwen2VLConfig.tiny_2_5()
    for f in ("hidden", "layers", "heads", "kv_heads", "intermediate", "vocab", "rms_eps", "rope_theta", "mrope_section",
              "v_depth", "v_embed", "v_heads", "v_mlp", "patch", "temporal", "merge", "vision_arch", "v_window", "v_fullatt",
              "image_token_id", "vision_start_id", "vision_end_id"):
        assert getattr(cfg, f) == getattr(tiny, f), f
    assert cfg.eos_ids[:2] == (503, 505)                                   # generation_config.json (config.json says 503)
    # preprocessor_config.json: NON-default bounds (size.shortest_edge / longest_edge spelling) and the CLIP constants
    assert (cfg.min_pixels, cfg.max_pixels) == (28 * 28 * 6, 28 * 28 * 30) != (tiny.min_pixels, tiny.max_pixels)
    assert cfg.image_mean == pytest.approx((0.48145466, 0.4578275, 0.40821073)) and cfg.image_std[0] == pytest.approx(0.26862954)
    cfg.validate_for_kernels()
    # the request through OUR tokenizer + chat layout == HF's chat template + processor expansion
    msgs = _messages(exp)
    img = decode_data_uri(msgs[0]["content"][1]["image_url"]["url"])
    th, tw = target_size(img.size, cfg.patch, cfg.merge, cfg.min_pixels, cfg.max_pixels)
    assert [1, th // cfg.patch, tw // cfg.patch] == exp["grid"][0].tolist()          # 140 x 140 here; the defaults would keep 196 x 168
    tok = HFTokenizer(d, cfg.image_token_id, cfg.vision_start_id, cfg.vision_end_id, cfg.eos_ids)
    ids = build_chat_ids(tok, msgs, [(th // cfg.patch) * (tw // cfg.patch) // cfg.merge ** 2])
    assert ids == exp["ids"].tolist()
    assert tok.decode(exp["tokens"].tolist()) == str(exp["reply"])
    # a preprocessor_config.json that contradicts config.json is refused, the older min_pixels / max_pixels spelling is read
    pre = json.load(open(os.path.join(d, "preprocessor_config.json")))
    json.dump({**pre, "size": None, "min_pixels": 3136, "max_pixels": 50176}, open(os.path.join(d, "preprocessor_config.json"), "w"))
    c2 = Qwen2VLConfig.from_hf_dir(d)
    assert (c2.min_pixels, c2.max_pixels) == (3136, 50176)
    json.dump({**pre, "patch_size": 16}, open(os.path.join(d, "preprocessor_config.json"), "w"))
    with pytest.raises(ValueError):
        Qwen2VLConfig.from_hf_dir(d)


@pytest.mark.parametrize("family", QWEN)
def test_qwen_dir_tensor_names_reach_the_packer(family, tmp_path, monkeypatch):
    """load_safetensors_dir reads the file save_pretrained's names describe and hands pack_device_weights every tensor it
    needs (the packing itself - bf16 kernel layouts - runs on the GPU tests; here the name normalisation is the point)."""
    from vision_inspection_system_amd import weights as W
    from vision_inspection_system_amd.config import Qwen2VLConfig
    d = materialize_hf_dir(family, tmp_path / family)
    cfg = Qwen2VLConfig.from_hf_dir(d)
    seen = {}
    monkeypatch.setattr(W, "pack_device_weights", lambda c, sd, dev: seen.update(sd) or "packed")
    assert W.load_safetensors_dir(cfg, d, "cpu") == "packed"
    normalised = {W._norm_key(k) for k in seen}
    assert set(W.tensor_shapes(cfg)) <= normalised
    ref = W.synth_state_dict(cfg, seed=0)
    for k, v in seen.items():
        assert torch.equal(v.reshape(-1).float(), ref[W._norm_key(k)].reshape(-1).float()), k
    # the names a transformers 5 module tree uses (model.language_model.*, model.visual.*) normalise to the same set
    assert W._norm_key("model.language_model.layers.0.mlp.up_proj.weight") == "model.layers.0.mlp.up_proj.weight"
    assert W._norm_key("model.visual.blocks.1.attn.qkv.bias") == "visual.blocks.1.attn.qkv.bias"


def test_mllama_dir_config_tokenizer_and_chat_ids(tmp_path, monkeypatch):
    from vision_inspection_system_amd import mllama_weights as MW
    from vision_inspection_system_amd.client import local_model_type
    from vision_inspection_system_amd.tokenizer import LlamaHFTokenizer, build_llama_chat_ids
    d = materialize_hf_dir("mllama_tiny", tmp_path / "mllama_tiny")
    exp = hf_expected("mllama_tiny")
    assert local_model_type(d) == "mllama"
    cfg = MW.config_from_hf_dir(d)
    tiny = MW.MllamaConfig.tiny()
    for f in ("hidden", "layers", "heads", "kv_heads", "intermediate", "vocab", "rms_eps", "rope_theta", "rope_factor",
              "rope_low_freq", "rope_high_freq", "rope_orig_ctx", "cross_layers", "image_token_id", "v_hidden", "v_heads",
              "v_layers", "v_global_layers", "v_mlp", "v_inter", "v_eps", "image_size", "patch", "max_tiles"):
        assert getattr(cfg, f) == getattr(tiny, f), f
    assert cfg.eos_ids == (501, 504)
    cfg.validate_for_kernels()
    tok = LlamaHFTokenizer(d, cfg.image_token_id, cfg.eos_ids)
    assert build_llama_chat_ids(tok, _messages(exp), 1) == exp["ids"].tolist()
    assert tok.decode(exp["tokens"].tolist()) == str(exp["reply"])
    seen = {}
    monkeypatch.setattr(MW, "pack_device_weights", lambda c, sd, dev: seen.update(sd) or "packed")
    assert MW.load_safetensors_dir(cfg, d, "cpu") == "packed"
    ref = MW.synth_state_dict(cfg, seed=0)
    assert set(seen) == set(ref)
    for k, v in seen.items():
        assert torch.equal(v.reshape(-1).float(), ref[k].reshape(-1).float()), k
    # a preprocessor_config.json with another tile geometry is refused
    pre = json.load(open(os.path.join(d, "preprocessor_config.json")))
    json.dump({**pre, "max_image_tiles": 2}, open(os.path.join(d, "preprocessor_config.json"), "w"))
    with pytest.raises(ValueError):
        MW.config_from_hf_dir(d)


def test_committed_dirs_are_small_and_carry_no_weights():
    for fam in QWEN + ["mllama_tiny"]:
        files = os.listdir(os.path.join(HF_DIRS, fam))
        assert "config.json" in files and "tokenizer.json" in files and "preprocessor_config.json" in files
        assert not any(f.endswith(".safetensors") or f.endswith(".bin") for f in files)
        assert sum(os.path.getsize(os.path.join(HF_DIRS, fam, f)) for f in files) < 400_000


# ----------------------------------------------------------------------------- GPU: through the client
@pytest.mark.gpu
@pytest.mark.parametrize("family", QWEN + ["mllama_tiny"])
def test_client_serves_a_model_directory_like_transformers(family, tmp_path, device, monkeypatch):
    """``chat.completions.create(model=<directory>)``: config, processor settings, tokenizer and safetensors all come from the
    directory; prompt ids equal HF's, the logits of every greedy step (HF's tokens teacher-forced) stay within the stated
    tolerance of the published model's, the reply text is HF's up to the first near-tie."""
    from helpers import teacher_forced_parity
    from vision_inspection_system_amd import client as CL
    d = materialize_hf_dir(family, tmp_path / family)
    exp = hf_expected(family)
    monkeypatch.setenv("VIS_MAX_BATCH", "2")
    monkeypatch.setenv("VIS_MAX_CTX", "512")
    monkeypatch.setenv("VIS_IGNORE_EOS", "1")
    try:
        cl = CL.LocalVLMClient(device=str(device))
        lm = CL.get_model(d, str(device))
        assert lm.family == ("mllama" if family == "mllama_tiny" else "qwen2_vl")
        msgs = _messages(exp)
        r = cl.chat.completions.create(model=d, messages=msgs, temperature=0.0, max_tokens=12)
        assert r.usage["prompt_tokens"] == len(exp["ids"]) and r.usage["completion_tokens"] == 12
        # logits, step by step, HF's tokens teacher-forced
        eng = lm.engine
        ref_logits = torch.from_numpy(exp["logits"])
        ref_toks = exp["tokens"].tolist()
        if family == "mllama_tiny":
            ids, frame = cl._prepare_mllama(lm, msgs)
            taps = {}
            eng.prefill(ids, CL._frame_to_device(frame, eng.device), taps=taps)
        else:
            from vision_inspection_system_amd import hip
            ids, frames = cl._prepare(lm, msgs)
            taps = {}
            eng.prefill(ids, [hip.resize_rgb(CL._frame_to_device(f, eng.device), th, tw) for f, (th, tw) in frames], taps=taps)
        assert list(ids) == exp["ids"].tolist()
        tol = 8e-2 if family == "mllama_tiny" else 6e-2
        ties = teacher_forced_parity(eng, taps["first_logits"], ref_toks, ref_logits, tol)
        # the free-running reply equals HF's up to the first near-tie (top-2 margin of the reference below 2 x tol)
        top2 = torch.topk(ref_logits, 2, dim=-1).values
        safe = int((((top2[:, 0] - top2[:, 1]) >= 2 * tol).long().cumprod(0)).sum())
        got = lm.tokenizer.encode(r.choices[0].message.content) if safe == 12 and ties == 0 else None
        if got is not None:
            assert r.choices[0].message.content == str(exp["reply"])
    finally:
        CL.drop_models()
