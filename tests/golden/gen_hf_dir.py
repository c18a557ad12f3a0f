"""DEV-ONLY generator of the real-model-directory fixtures (reference surface: config/models.yaml:5-18,
utils/config.py:42-76 - ``VLM_INSPECTOR_MODEL`` / ``model_id`` may name a model directory).

For each family the local engines serve - qwen2_vl, qwen2_5_vl, mllama - this script lets transformers 5.15.0
``save_pretrained`` a tiny seeded model and records what a real checkpoint directory looks like and what the published
model answers for one chat request:

  tests/golden/hf_dirs/<family>/config.json               exactly as transformers writes it (nested text_config / vision_config)
                               generation_config.json    ditto
                               preprocessor_config.json  the image processor's own file - with NON-default min / max pixels
                               tokenizer.json            a small byte-level BPE vocabulary (trained here with ``tokenizers`` on
                                                         own-worded text) carrying the Qwen / Llama-3 chat specials at the ids
                                                         the tiny configs use
                               tokenizer_config.json, chat_template.jinja   what PreTrainedTokenizerFast.save_pretrained writes
                               manifest.json             tensor names + shapes + dtype of model.safetensors AS WRITTEN by
                                                         save_pretrained (the 26 MB file itself is not committed: the weights
                                                         are the seeded ``synth_state_dict`` and tests/helpers.py rewrites the
                                                         file under exactly these names)
                               expected.npz              the request (image, text), the ids of HF's chat template + processor,
                                                         pixel grid, logits of 12 teacher-forced greedy steps, the tokens and
                                                         the decoded reply

The chat templates are transcribed from the published Qwen2-VL / Llama-3.2-Vision templates (no hub access here); the
request is the reference's shape: one user message, text part first, then the image (vlm_inspector.py:462-470).

Usage:  python tests/golden/gen_hf_dir.py
"""
import io
import json
import os
import shutil
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, HERE)

OUT = os.path.join(HERE, "hf_dirs")

QWEN_TEMPLATE = (
    "{% set image_count = namespace(value=0) %}{% for message in messages %}"
    "{% if loop.first and message['role'] != 'system' %}<|im_start|>system\nYou are a helpful assistant.<|im_end|>\n{% endif %}"
    "<|im_start|>{{ message['role'] }}\n"
    "{% if message['content'] is string %}{{ message['content'] }}<|im_end|>\n"
    "{% else %}{% for content in message['content'] %}"
    "{% if content['type'] == 'image' or 'image' in content or 'image_url' in content %}"
    "{% set image_count.value = image_count.value + 1 %}{% if add_vision_id %}Picture {{ image_count.value }}: {% endif %}"
    "<|vision_start|><|image_pad|><|vision_end|>"
    "{% elif 'text' in content %}{{ content['text'] }}{% endif %}{% endfor %}<|im_end|>\n{% endif %}{% endfor %}"
    "{% if add_generation_prompt %}<|im_start|>assistant\n{% endif %}")

LLAMA_TEMPLATE = (
    "{{- bos_token }}{%- for message in messages %}"
    "{{- '<|start_header_id|>' + message['role'] + '<|end_header_id|>\n\n' }}"
    "{%- if message['content'] is string %}{{- message['content'] | trim }}"
    "{%- else %}{%- for content in message['content'] %}"
    "{%- if content['type'] == 'image' %}{{- '<|image|>' }}"
    "{%- elif content['type'] == 'text' %}{{- content['text'] | trim }}{%- endif %}{%- endfor %}{%- endif %}"
    "{{- '<|eot_id|>' }}{%- endfor %}"
    "{%- if add_generation_prompt %}{{- '<|start_header_id|>assistant<|end_header_id|>\n\n' }}{%- endif %}")

CORPUS = [
    "You are a visual inspection assistant. Inspect the part in the image for defects and answer in JSON only.",
    "Report every crack, dent, scratch, corrosion spot, missing fastener or misalignment you can see, with its location.",
    '{"object_identified": "steel bracket", "overall_condition": "damaged", "defects": [{"type": "crack", "location": '
    '"upper left weld seam", "safety_impact": "CRITICAL", "confidence": "high", "recommended_action": "replace"}], '
    '"overall_confidence": "high", "analysis_reasoning": "one clear crack at the weld"}',
    "Criticality: medium. Domain: general. User notes: None provided. Bounding boxes are percentages of the image size.",
    "If the image is clean say so with high confidence; if you are unsure say uncertain and explain what blocks the view.",
    "system user assistant You are a helpful assistant. 0123456789 the quick brown fox jumps over the lazy dog",
] * 8

PROMPT = ("Inspect the part in the image for defects and answer in JSON only. Criticality: medium. Domain: general. "
          "Report every crack, dent or scratch with its location.")


def train_bpe(n_vocab: int):
    from tokenizers import Tokenizer, decoders, models, pre_tokenizers, trainers
    tok = Tokenizer(models.BPE())
    tok.pre_tokenizer = pre_tokenizers.ByteLevel(add_prefix_space=False)
    tok.decoder = decoders.ByteLevel()
    trainer = trainers.BpeTrainer(vocab_size=n_vocab, initial_alphabet=pre_tokenizers.ByteLevel.alphabet(),
                                  special_tokens=[], show_progress=False)
    tok.train_from_iterator(CORPUS, trainer)
    if tok.get_vocab_size() != n_vocab:
        raise RuntimeError(f"BPE training gave {tok.get_vocab_size()} tokens, wanted {n_vocab}: enlarge the corpus")
    return tok


def add_specials(tok, names):
    from tokenizers import AddedToken
    tok.add_special_tokens([AddedToken(n, special=True, normalized=False) for n in names])
    return {n: tok.token_to_id(n) for n in names}


def png_data_uri(frame: np.ndarray) -> bytes:
    from PIL import Image
    b = io.BytesIO()
    Image.fromarray(frame).save(b, format="PNG")
    return b.getvalue()


def manifest_of(path):
    from safetensors import safe_open
    out = {}
    with safe_open(os.path.join(path, "model.safetensors"), "pt") as f:
        for k in f.keys():
            t = f.get_tensor(k)
            out[k] = {"shape": list(t.shape), "dtype": str(t.dtype).replace("torch.", "")}
    return out


def finish_dir(d, model):
    """save_pretrained, record the manifest, drop the big file."""
    model.save_pretrained(d, safe_serialization=True)
    with open(os.path.join(d, "manifest.json"), "w") as f:
        json.dump(manifest_of(d), f, indent=0, sort_keys=True)
    os.remove(os.path.join(d, "model.safetensors"))


def greedy_steps(step_fn, first_logits, n_new):
    """teacher-forced greedy loop on the HF model: step_fn(token) -> next logits"""
    toks, logits = [], [first_logits]
    cur = int(torch.argmax(first_logits))
    toks.append(cur)
    for _ in range(1, n_new):
        lg = step_fn(cur)
        logits.append(lg)
        cur = int(torch.argmax(lg))
        toks.append(cur)
    return toks, torch.stack(logits)


def qwen_family(name, cfg, build_hf_model, synth_state_dict):
    from transformers import PreTrainedTokenizerFast
    from transformers.models.qwen2_vl.image_processing_pil_qwen2_vl import Qwen2VLImageProcessorPil
    d = os.path.join(OUT, name)
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    # ---- vocabulary: 500 byte-level BPE tokens, then the chat specials at the ids the tiny config declares
    tok = train_bpe(500)
    ids = add_specials(tok, ["<|image_pad|>", "<|vision_start|>", "<|vision_end|>", "<|im_end|>", "<|im_start|>",
                             "<|endoftext|>", "<|vision_pad|>", "<|video_pad|>"])
    assert (ids["<|image_pad|>"], ids["<|vision_start|>"], ids["<|vision_end|>"], ids["<|im_end|>"]) == \
        (cfg.image_token_id, cfg.vision_start_id, cfg.vision_end_id, cfg.eos_ids[0]), ids
    tok.save(os.path.join(d, "tokenizer.json"))
    fast = PreTrainedTokenizerFast(tokenizer_file=os.path.join(d, "tokenizer.json"), eos_token="<|im_end|>",
                                   pad_token="<|endoftext|>", chat_template=QWEN_TEMPLATE)
    fast.save_pretrained(d)
    # ---- model + config exactly as save_pretrained writes them (eos ids as a released checkpoint carries them)
    model = build_hf_model(cfg, synth_state_dict(cfg, seed=0))
    model.config.eos_token_id = cfg.eos_ids[0]
    model.config.text_config.eos_token_id = cfg.eos_ids[0]
    model.generation_config.eos_token_id = [cfg.eos_ids[0], ids["<|endoftext|>"]]
    finish_dir(d, model)
    # ---- image processor with NON-default pixel bounds: a 200 x 180 frame must come out smaller than the default would make it
    proc = Qwen2VLImageProcessorPil(size={"shortest_edge": 28 * 28 * 6, "longest_edge": 28 * 28 * 30})
    proc.save_pretrained(d)
    rng = np.random.default_rng(77)
    frame = rng.integers(0, 256, (200, 180, 3), dtype=np.uint8)
    from PIL import Image
    feats = proc(images=[Image.fromarray(frame)], return_tensors="pt")
    pv, grid = feats["pixel_values"].float(), feats["image_grid_thw"]
    n_img = int(grid[0].prod()) // cfg.merge ** 2
    # ---- the request through HF's template: text part first, then the image (the reference's order)
    messages = [{"role": "user", "content": [{"type": "text", "text": PROMPT}, {"type": "image"}]}]
    text = fast.apply_chat_template(messages, add_generation_prompt=True, tokenize=False)
    text = text.replace("<|image_pad|>", "<|image_pad|>" * n_img)          # what Qwen2VLProcessor does with the grid
    prompt_ids = fast(text, add_special_tokens=False)["input_ids"]
    input_ids = torch.tensor([prompt_ids], dtype=torch.long)
    mm = (input_ids == cfg.image_token_id).int()
    n_new = 12
    with torch.no_grad():
        o = model(input_ids=input_ids, pixel_values=pv, image_grid_thw=grid, mm_token_type_ids=mm, use_cache=True)
        state = {"past": o.past_key_values, "n": input_ids.shape[1]}
        pos, deltas = model.model.get_rope_index(input_ids, mm, image_grid_thw=grid)

        def step(tokid):
            p = (torch.full((3, 1, 1), state["n"]) + deltas.view(1, 1, 1)).long()
            oo = model(input_ids=torch.tensor([[tokid]]), past_key_values=state["past"], position_ids=p, use_cache=True)
            state["past"], state["n"] = oo.past_key_values, state["n"] + 1
            return oo.logits[0, -1].float()
        toks, logits = greedy_steps(step, o.logits[0, -1].float(), n_new)
        # cross-check the hand-rolled cache loop against HF's own generate
        gen = model.generate(input_ids=input_ids, pixel_values=pv, image_grid_thw=grid, mm_token_type_ids=mm,
                             max_new_tokens=n_new, do_sample=False, eos_token_id=None)
        assert gen[0, len(prompt_ids):].tolist() == toks, (gen[0, len(prompt_ids):].tolist(), toks)
    np.savez_compressed(os.path.join(d, "expected.npz"), png=np.frombuffer(png_data_uri(frame), dtype=np.uint8),
                        prompt=np.array(PROMPT), chat_text=np.array(text), ids=np.array(prompt_ids), grid=grid.numpy(),
                        logits=logits.numpy().astype(np.float32), tokens=np.array(toks),
                        reply=np.array(fast.decode(toks, skip_special_tokens=True)))
    print(name, "prompt ids", len(prompt_ids), "image tokens", n_img, "grid", grid.tolist(), "tokens", toks)


def mllama_family():
    from transformers import PreTrainedTokenizerFast
    from transformers.models.mllama.image_processing_pil_mllama import MllamaImageProcessorPil
    import gen_mllama_golden as GM
    from vision_inspection_system_amd.mllama_weights import MllamaConfig, synth_state_dict
    cfg = MllamaConfig.tiny()
    d = os.path.join(OUT, "mllama_tiny")
    shutil.rmtree(d, ignore_errors=True)
    os.makedirs(d)
    tok = train_bpe(500)
    names = ["<|begin_of_text|>", "<|end_of_text|>", "<|start_header_id|>", "<|end_header_id|>", "<|eot_id|>",
             "<|finetune_right_pad_id|>"] + [f"<|reserved_special_token_{i}|>" for i in range(6)] + ["<|image|>"]
    ids = add_specials(tok, names)
    assert ids["<|image|>"] == cfg.image_token_id == 512, ids
    tok.save(os.path.join(d, "tokenizer.json"))
    fast = PreTrainedTokenizerFast(tokenizer_file=os.path.join(d, "tokenizer.json"), bos_token="<|begin_of_text|>",
                                   eos_token="<|eot_id|>", pad_token="<|finetune_right_pad_id|>", chat_template=LLAMA_TEMPLATE)
    fast.save_pretrained(d)
    model = GM.build_hf_model(cfg, synth_state_dict(cfg, seed=0))
    eos = [ids["<|end_of_text|>"], ids["<|eot_id|>"]]
    model.config.text_config.eos_token_id = eos
    model.config.text_config.bos_token_id = ids["<|begin_of_text|>"]
    model.config.text_config.pad_token_id = ids["<|finetune_right_pad_id|>"]
    model.generation_config.eos_token_id = eos
    finish_dir(d, model)
    proc = MllamaImageProcessorPil(size={"height": cfg.image_size, "width": cfg.image_size}, max_image_tiles=cfg.max_tiles,
                                   image_mean=GM.CLIP_MEAN, image_std=GM.CLIP_STD)
    proc.save_pretrained(d)
    rng = np.random.default_rng(78)
    frame = rng.integers(0, 256, (100, 90, 3), dtype=np.uint8)
    messages = [{"role": "user", "content": [{"type": "text", "text": PROMPT}, {"type": "image"}]}]
    text = fast.apply_chat_template(messages, add_generation_prompt=True, tokenize=False)
    prompt_ids = fast(text, add_special_tokens=False)["input_ids"]
    enc = proc.preprocess([[frame]], return_tensors="pt")
    pv = enc["pixel_values"].float()
    n_tiles = int(enc["num_tiles"][0][0])
    S, n_new = len(prompt_ids), 12
    loc = prompt_ids.index(cfg.image_token_id)
    xmask = torch.zeros(1, S + n_new, 1, cfg.max_tiles)
    xmask[0, loc:, 0, :n_tiles] = 1
    with torch.no_grad():
        o = model(input_ids=torch.tensor([prompt_ids]), pixel_values=pv, aspect_ratio_ids=enc["aspect_ratio_ids"],
                  aspect_ratio_mask=enc["aspect_ratio_mask"], cross_attention_mask=xmask[:, :S], use_cache=True)
        state = {"past": o.past_key_values, "n": S}

        def step(tokid):
            state["n"] += 1
            oo = model(input_ids=torch.tensor([[tokid]]), past_key_values=state["past"],
                       cross_attention_mask=xmask[:, :state["n"]], use_cache=True)
            state["past"] = oo.past_key_values
            return oo.logits[0, -1].float()
        toks, logits = greedy_steps(step, o.logits[0, -1].float(), n_new)
    np.savez_compressed(os.path.join(d, "expected.npz"), png=np.frombuffer(png_data_uri(frame), dtype=np.uint8),
                        prompt=np.array(PROMPT), chat_text=np.array(text), ids=np.array(prompt_ids),
                        n_tiles=np.array(n_tiles), logits=logits.numpy().astype(np.float32), tokens=np.array(toks),
                        reply=np.array(fast.decode(toks, skip_special_tokens=True)))
    print("mllama_tiny prompt ids", S, "tiles", n_tiles, "tokens", toks)


def main():
    from vision_inspection_system_amd.config import Qwen2VLConfig
    from vision_inspection_system_amd.weights import synth_state_dict
    import gen_qwen2vl_golden as G2
    import gen_qwen25vl_golden as G25
    os.makedirs(OUT, exist_ok=True)
    qwen_family("qwen2vl_tiny", Qwen2VLConfig.tiny(), G2.build_hf_model, synth_state_dict)
    qwen_family("qwen25vl_tiny", Qwen2VLConfig.tiny_2_5(), G25.build_hf_model, synth_state_dict)
    mllama_family()


if __name__ == "__main__":
    main()
