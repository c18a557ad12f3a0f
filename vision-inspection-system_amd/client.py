"""Boundary B1: a ``chat.completions.create``-shaped client backed by the local MI355X engine.

The reference talks to its models through exactly one call shape,
    client.chat.completions.create(model=, messages=, temperature=, max_tokens=).choices[0].message.content
(``huggingface_hub.InferenceClient`` at src/agents/vlm_inspector.py:32,:105-111 and
src/agents/vlm_auditor.py:152-158; ``groq.Groq`` at :117-129; text-only health check
vlm_inspector.py:533-544).  ``LocalVLMClient`` offers the same attribute chain and return shape, so
the reference's agents can be pointed at it without touching their code.

Errors: ordinary exceptions whose messages never contain "429", "rate", "413" or "payload" - the
substrings the reference's retry logic keys on (vlm_inspector.py:113-140).
"""
from __future__ import annotations

import logging
import os
import threading
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Tuple

from .config import LOCAL_PROVIDER, Qwen2VLConfig

logger = logging.getLogger("vision_inspection_system_amd.client")


# ----------------------------------------------------------------------------- response objects
@dataclass
class _Message:
    content: str
    role: str = "assistant"


@dataclass
class _Choice:
    message: _Message
    index: int = 0
    finish_reason: str = "stop"


@dataclass
class ChatCompletion:
    choices: List[_Choice]
    model: str = ""
    usage: Dict[str, int] = field(default_factory=dict)
    # device time per stage of the batch this reply was part of (extension; the reference logs wall time only:
    # vlm_inspector.py:473-479): {"prefill_ms", "decode_ms", "decode_steps", "sequences", "prompt_tokens"}
    timings: Dict[str, float] = field(default_factory=dict)


class _Completions:
    def __init__(self, owner):
        self._owner = owner

    def create(self, model: Optional[str] = None, messages: Optional[list] = None,
               temperature: Optional[float] = None, max_tokens: Optional[int] = None, **kwargs) -> ChatCompletion:
        return self._owner._complete(model, messages or [], temperature, max_tokens, **kwargs)


class _Chat:
    def __init__(self, owner):
        self.completions = _Completions(owner)


# device time of every generate_batch group served in this process (extension; bench.py --workload batch256 reads and
# clears it): [{"model", "prefill_ms", "decode_ms", "decode_steps", "sequences", "prompt_tokens"}]
TIMING_LOG: List[dict] = []

# ----------------------------------------------------------------------------- engine registry
_ENGINES: Dict[Tuple[str, str], Any] = {}
_ENGINES_LOCK = threading.Lock()


@dataclass
class LoadedModel:
    engine: Any
    tokenizer: Any
    cfg: Any
    model_id: str
    family: str = "qwen2_vl"      # or "mllama" (row f2: the Auditor's Llama-3.2-11B-Vision fallback)


def resolve_model_dir(model_id: str) -> Optional[str]:
    """A hub-style name is only ever mapped to a LOCAL directory: ``$VIS_MODEL_ROOT/<name>`` or
    ``$VIS_MODEL_ROOT/<org>--<name>``.  Nothing is downloaded."""
    if os.path.isdir(model_id):
        return model_id
    root = os.environ.get("VIS_MODEL_ROOT")
    if root:
        for cand in (os.path.join(root, model_id), os.path.join(root, model_id.replace("/", "--")),
                     os.path.join(root, model_id.split("/")[-1])):
            if os.path.isdir(cand):
                return cand
    return None


def get_model(model_id: str, device: Optional[str] = None) -> LoadedModel:
    """Process-wide singleton per (model, device): the reference constructs a NEW agent object on every
    node call (src/orchestration/nodes.py:128,:230), so the 16.6 GB model must not live in the agent."""
    import torch
    from .engine import Qwen2VLEngine
    from .tokenizer import ByteTokenizer, HFTokenizer
    from . import weights as W
    if device is None:
        with _ENGINES_LOCK:      # a model that is already loaded / registered on exactly one device: no device query needed
            hits = [k for k in _ENGINES if k[0] == model_id]
            if len(hits) == 1:
                return _ENGINES[hits[0]]
        device = f"cuda:{torch.cuda.current_device()}" if torch.cuda.is_available() else "cuda:0"
    key = (model_id, str(device))
    with _ENGINES_LOCK:
        if key in _ENGINES:
            return _ENGINES[key]
        # KV-cache rows per sequence.  Default: the reference's request fits as the reference sends it - ~2300 prompt tokens
        # (inspection prompt + one 1024 x 1024 image) + its own default max_tokens = 2048 (/root/reference utils/config.py:50-53)
        # = 4348 -> 4608 (72 context splits: the chained decode launch stays resident up to ~6700)
        max_ctx = int(os.environ.get("VIS_MAX_CTX", "4608"))
        max_batch = max(1, min(64, int(os.environ.get("VIS_MAX_BATCH", "64"))))
        mllama = _load_mllama(model_id, device, max_ctx, max_batch)
        if mllama is not None:
            _ENGINES[key] = mllama
            return mllama
        if model_id.startswith("synthetic:"):
            parts = model_id.split(":")
            kind = parts[1]
            seed = int(parts[2]) if len(parts) > 2 else 0
            if kind == "tiny":
                cfg = Qwen2VLConfig.tiny()
                w = W.pack_device_weights(cfg, W.synth_state_dict(cfg, seed), device)
                max_ctx = min(max_ctx, int(os.environ.get("VIS_TINY_MAX_CTX", "1024")))
            elif kind in ("7b", "qwen2-vl-7b"):
                cfg = Qwen2VLConfig.qwen2_vl_7b()
                w = W.random_device_weights(cfg, device, seed)
            elif kind in ("tiny25", "qwen2.5-vl-tiny"):
                cfg = Qwen2VLConfig.tiny_2_5()
                w = W.pack_device_weights(cfg, W.synth_state_dict(cfg, seed), device)
                max_ctx = min(max_ctx, int(os.environ.get("VIS_TINY_MAX_CTX", "1024")))
            elif kind in ("7b25", "qwen2.5-vl-7b"):
                cfg = Qwen2VLConfig.qwen2_5_vl_7b()
                w = W.random_device_weights(cfg, device, seed)
            else:
                raise ValueError(f"unknown synthetic model {kind!r} (use synthetic:tiny, synthetic:7b, synthetic:tiny25 "
                                 f"or synthetic:qwen2.5-vl-7b)")
            tok = ByteTokenizer(cfg.vocab, cfg.image_token_id, cfg.vision_start_id, cfg.vision_end_id, cfg.eos_ids)
        else:
            path = resolve_model_dir(model_id)
            if path is None:
                raise FileNotFoundError(
                    f"model {model_id!r} is not a local directory and VIS_MODEL_ROOT has no copy of it; the "
                    f"'{LOCAL_PROVIDER}' provider only loads local files (config.json, *.safetensors, tokenizer.json)")
            local_model_type(path)          # refuses anything but qwen2_vl / qwen2_5_vl (mllama was handled above)
            cfg = Qwen2VLConfig.from_hf_dir(path)
            w = W.load_safetensors_dir(cfg, path, device)
            tok = HFTokenizer(path, cfg.image_token_id, cfg.vision_start_id, cfg.vision_end_id, cfg.eos_ids)
        lm = LoadedModel(Qwen2VLEngine(cfg, w, device, max_ctx=max_ctx, max_batch=max_batch,
                                       decode_weights=os.environ.get("VIS_DECODE_WEIGHTS", "bf16"),
                                       prefill_dtype=os.environ.get("VIS_PREFILL_DTYPE", "bf16")), tok, cfg, model_id)
        _ENGINES[key] = lm
        return lm


def _load_mllama(model_id: str, device, max_ctx: int, max_batch: int = 1) -> Optional[LoadedModel]:
    """mllama family (synthetic:mllama-tiny[:seed], synthetic:mllama-11b, or a local directory whose config.json
    says model_type "mllama"); None when ``model_id`` is not an mllama model."""
    import json
    from . import mllama_weights as MW
    from .mllama_engine import MllamaEngine
    from .tokenizer import LlamaByteTokenizer, LlamaHFTokenizer
    if model_id.startswith("synthetic:mllama"):
        parts = model_id.split(":")
        seed = int(parts[2]) if len(parts) > 2 else 0
        if parts[1] == "mllama-tiny":
            cfg = MW.MllamaConfig.tiny()
            w = MW.pack_device_weights(cfg, MW.synth_state_dict(cfg, seed), device)
            max_ctx = min(max_ctx, int(os.environ.get("VIS_TINY_MAX_CTX", "1024")))
        elif parts[1] in ("mllama-11b", "mllama"):
            cfg = MW.MllamaConfig.mllama_11b()
            w = MW.random_device_weights(cfg, device, seed)
        else:
            raise ValueError(f"unknown synthetic model {parts[1]!r}")
        tok = LlamaByteTokenizer(cfg.vocab, cfg.image_token_id, cfg.eos_ids)
        return LoadedModel(MllamaEngine(cfg, w, device, max_ctx=max_ctx, max_batch=max_batch), tok, cfg, model_id, "mllama")
    path = resolve_model_dir(model_id)
    if path is None or not os.path.exists(os.path.join(path, "config.json")):
        return None
    with open(os.path.join(path, "config.json")) as f:
        if json.load(f).get("model_type") != "mllama":
            return None
    cfg = MW.config_from_hf_dir(path)
    w = MW.load_safetensors_dir(cfg, path, device)
    tok = LlamaHFTokenizer(path, cfg.image_token_id, cfg.eos_ids)
    return LoadedModel(MllamaEngine(cfg, w, device, max_ctx=max_ctx, max_batch=max_batch), tok, cfg, model_id, "mllama")


def _reply_text(model_id: str, decoded: str) -> str:
    """Throughput runs on ``synthetic:`` (seeded random) weights generate noise, which would send every image down the
    agents' failure + retry path (nodes.py retries with back-off) and time THAT instead of the pipeline.  For synthetic
    models only, VIS_SYNTHETIC_REPLY substitutes a fixed reply text AFTER the full generation has run (tools/ingest_bench.py);
    real checkpoints are never affected."""
    if model_id.startswith("synthetic:"):
        fixed = os.environ.get("VIS_SYNTHETIC_REPLY")
        if fixed:
            return fixed
    return decoded


def drop_models() -> None:
    with _ENGINES_LOCK:
        _ENGINES.clear()


def register_model(model_id: str, device: str, lm: LoadedModel) -> None:
    """Serve ``model_id`` on ``device`` from an engine the caller already built (bench.py times the client on the
    engine it has just measured instead of loading a second 16.6 GB replica)."""
    with _ENGINES_LOCK:
        _ENGINES[(model_id, str(device))] = lm


def unregister_model(model_id: str, device: str) -> None:
    with _ENGINES_LOCK:
        _ENGINES.pop((model_id, str(device)), None)


SUPPORTED_MODEL_TYPES = ("qwen2_vl", "qwen2_5_vl", "mllama")


def local_model_type(path: str) -> str:
    """``model_type`` of a local HuggingFace directory, checked against what the engines implement.  The
    reference's code default for both agents is Qwen/Qwen2.5-VL-7B-Instruct (utils/config.py:42-45,:59-64), its
    README names Qwen2-VL-7B and Llama-3.2-11B-Vision; anything else is refused HERE with a clear message instead of
    failing with a KeyError deep inside a weight loader."""
    import json
    cfg_path = os.path.join(path, "config.json")
    if not os.path.exists(cfg_path):
        raise FileNotFoundError(f"{cfg_path} not found: a local model directory needs config.json, *.safetensors and "
                                f"tokenizer.json")
    with open(cfg_path) as f:
        mt = json.load(f).get("model_type")
    if mt not in SUPPORTED_MODEL_TYPES:
        raise ValueError(f"model directory {path!r} has model_type {mt!r}; the '{LOCAL_PROVIDER}' provider serves "
                         f"{', '.join(SUPPORTED_MODEL_TYPES)} only")
    return mt


def _frame_to_device(f, device):
    """Host result of a request's decode -> uint8 [H, W, 3] device frame: either decoded pixels (PIL path, upload) or the
    entropy-decoded JPEG (upload of the coefficients + the IDCT / upsampling / colour kernels)."""
    import torch
    from . import jpeg
    if isinstance(f, jpeg.JpegCoeffs):
        return jpeg.to_rgb_device(f, device)
    from . import hip
    return hip.upload(f, device)


# ----------------------------------------------------------------------------- clients
class LocalVLMClient:
    """``InferenceClient``-shaped facade over the MI355X engine."""

    accepts_futures = True      # complete_many takes Futures of messages (agents.prepare_many) and streams them in

    def __init__(self, api_key: Optional[str] = None, device: Optional[str] = None, default_model: Optional[str] = None,
                 seed: int = 0, **_ignored):
        self.device = device
        self.default_model = default_model
        self.seed = seed
        self.chat = _Chat(self)

    def _prepare(self, lm, messages):
        """messages -> (token ids, [(decoded uint8 RGB frame, (target_h, target_w))]) for one request.
        The JPEG is decoded on the host; the bicubic resample to the smart_resize target runs on the GPU
        (hip.resize_rgb, bit-exact with PIL) unless VIS_GPU_RESIZE=0 asks for the host PIL path."""
        import numpy as np
        from . import jpeg
        from .image_processing import decode_data_uri, resize_for_model, target_size
        from .tokenizer import build_chat_ids
        cfg = lm.cfg
        gpu_resize = os.environ.get("VIS_GPU_RESIZE", "1") != "0"
        frames = []
        for m in messages:
            content = m.get("content")
            if isinstance(content, list):
                for part in content:
                    if part.get("type") == "image_url":
                        url = part["image_url"]["url"] if isinstance(part.get("image_url"), dict) else part["image_url"]
                        # baseline JPEG (what the reference's agents send): Huffman decode here, on this pool thread;
                        # IDCT / upsampling / colour conversion on the GPU (jpeg.py).  Other flavours: PIL.
                        jc = jpeg.parse_data_uri(url) if (gpu_resize and jpeg.enabled()) else None
                        if jc is not None:
                            frames.append((jc, target_size(jc.size, cfg.patch, cfg.merge, cfg.min_pixels, cfg.max_pixels)))
                            continue
                        img = decode_data_uri(url)
                        th, tw = target_size(img.size, cfg.patch, cfg.merge, cfg.min_pixels, cfg.max_pixels)
                        if gpu_resize:
                            frames.append((np.array(img, dtype=np.uint8), (th, tw)))
                        else:
                            frames.append((resize_for_model(img, cfg.patch, cfg.merge, cfg.min_pixels,
                                                            cfg.max_pixels), (th, tw)))
        counts = [(th // cfg.patch) * (tw // cfg.patch) // cfg.merge ** 2 for _, (th, tw) in frames]
        return build_chat_ids(lm.tokenizer, messages, counts), frames

    def _complete(self, model, messages, temperature, max_tokens, **kwargs) -> ChatCompletion:
        return self.complete_many(model, [messages], temperature, max_tokens)[0]

    def complete_many(self, model, batch_of_messages, temperature=None, max_tokens=None) -> List[ChatCompletion]:
        """Several independent requests in one go: per-request prefill, then ONE shared decode loop in which every
        weight is streamed once per step for all of them (engine.generate_batch).  Groups larger than the
        engine's max_batch are processed in consecutive chunks.  Extension of the reference's call shape used by
        the batch path; ``chat.completions.create`` is the single-request form of it."""
        import torch
        model_id = model or self.default_model
        if not model_id:
            raise ValueError("no model given")
        lm = get_model(model_id, self.device)
        eng, tok = lm.engine, lm.tokenizer
        max_new = int(max_tokens) if max_tokens else 512
        temp = float(temperature) if temperature else 0.0
        out: List[ChatCompletion] = []
        if lm.family == "mllama":
            return self._complete_mllama_many(lm, batch_of_messages, temp, max_new)
        # Service-side decode (base64 + JPEG) of every request on the ingest pool.  A request may arrive as a Future of
        # its messages (the agents' prepare_many: the request-side encode is still running on the same pool); its decode
        # is queued the moment that encode finishes, ahead of the encodes still waiting (ingest.then).  The engine receives the
        # requests as callables and resolves them in order, so its first prompt pass starts as soon as image 0 is
        # decoded, and group i+1 decodes while group i is in its decode loop.
        # Eager requests (plain message lists): a request that fails to decode fails the call, like a malformed request
        # to the service.  Future requests: the failure (encode or decode) stays that request's own - its place in the
        # returned list holds the exception.
        from concurrent.futures import Future
        from . import hip, ingest
        lazy = any(isinstance(m, Future) for m in batch_of_messages)

        def prepare(msgs):
            with ingest.span("service-side decode (base64 + Huffman, pool thread)"):
                return self._prepare(lm, msgs)

        futs = [ingest.then(m, prepare) for m in batch_of_messages]
        n_ids = {}

        def resolver(j):
            def resolve():
                with ingest.span("engine thread: waiting for a request's encode + decode"):
                    ids, frames = futs[j].result()
                n_ids[j] = len(ids)
                if getattr(eng, "host_only", False):      # bench.py --dry-ingest: an engine stand-in that measures the host side
                    return ids, frames
                with ingest.span("engine thread: H2D of coefficients + IDCT / resize launches"):
                    return ids, [hip.resize_rgb(_frame_to_device(f, eng.device), th, tw) for f, (th, tw) in frames]
            return resolve

        with eng.lock:
            for i in range(0, len(futs), eng.max_batch):
                idx = range(i, min(len(futs), i + eng.max_batch))
                toks = eng.generate_batch([resolver(j) for j in idx], max_new_tokens=max_new, temperature=temp, seed=self.seed,
                                          ignore_eos=os.environ.get("VIS_IGNORE_EOS") == "1")
                timing = dict(getattr(eng, "last_timing", {}))
                if timing:
                    TIMING_LOG.append({"model": model_id, **timing})
                    del TIMING_LOG[:-4096]
                    logger.debug("%s: %d request(s): prompt pass %.1f ms, %d decode steps in %.1f ms (device time)", model_id,
                                 len(idx), timing["prefill_ms"], timing["decode_steps"], timing["decode_ms"])
                for j, t in zip(idx, toks):
                    if isinstance(t, Exception):
                        if not lazy:
                            raise t
                        out.append(t)
                        continue
                    out.append(ChatCompletion([_Choice(_Message(_reply_text(model_id, tok.decode(t))))], model=model_id,
                                              usage={"prompt_tokens": n_ids[j], "completion_tokens": len(t),
                                                     "total_tokens": n_ids[j] + len(t)}, timings=timing))
        return out


    def _prepare_mllama(self, lm, messages):
        """messages -> (token ids, decoded uint8 RGB frame or None).  The JPEG is decoded on the host, the tile canvas is
        chosen on the host; bilinear resample / normalise / patchify and everything after run on the GPU."""
        import numpy as np
        from . import jpeg
        from .image_processing import decode_data_uri
        from .tokenizer import build_llama_chat_ids
        frames = []
        for m in messages:
            content = m.get("content")
            if isinstance(content, list):
                for part in content:
                    if part.get("type") == "image_url":
                        url = part["image_url"]["url"] if isinstance(part.get("image_url"), dict) else part["image_url"]
                        jc = jpeg.parse_data_uri(url) if jpeg.enabled() else None
                        frames.append(jc if jc is not None else np.array(decode_data_uri(url), dtype=np.uint8))
        if len(frames) > 1:
            raise ValueError("the mllama backend takes one image per request (what the reference sends)")
        return build_llama_chat_ids(lm.tokenizer, messages, len(frames)), (frames[0] if frames else None)

    def _complete_mllama_many(self, lm, batch_of_messages, temp: float, max_new: int) -> List[ChatCompletion]:
        """Requests with an image share ONE decode loop in groups of the engine's max_batch (MllamaEngine.generate_batch:
        per-request prompt pass, weights streamed once per generated token for the whole group); text-only requests
        (the agents' health check) take the single-sequence path."""
        import torch
        from concurrent.futures import Future
        eng, tok = lm.engine, lm.tokenizer
        from . import ingest
        ignore_eos = os.environ.get("VIS_IGNORE_EOS") == "1"

        def completion(n_ids, t):
            return ChatCompletion([_Choice(_Message(_reply_text(lm.model_id, tok.decode(t))))], model=lm.model_id,
                                  usage={"prompt_tokens": n_ids, "completion_tokens": len(t), "total_tokens": n_ids + len(t)},
                                  timings=dict(getattr(eng, "last_timing", {})))

        futs = [ingest.then(m, lambda msgs: self._prepare_mllama(lm, msgs)) for m in batch_of_messages]
        if any(isinstance(m, Future) for m in batch_of_messages):
            # the batch seam (verify_many): every request carries an image; requests are resolved in order by the engine
            # while it already runs the earlier prompt passes; a failed request keeps its exception as its result
            out: list = []
            n_ids = {}

            def resolver(j):
                def resolve():
                    ids, f = futs[j].result()
                    n_ids[j] = len(ids)
                    return ids, (_frame_to_device(f, eng.device) if f is not None else None)
                return resolve

            with eng.lock:
                for g0 in range(0, len(futs), eng.max_batch):
                    idx = range(g0, min(len(futs), g0 + eng.max_batch))
                    outs = eng.generate_batch([resolver(j) for j in idx], max_new_tokens=max_new, temperature=temp,
                                              seed=self.seed, stop_on_eos=not ignore_eos)
                    if getattr(eng, "last_timing", None):
                        TIMING_LOG.append({"model": lm.model_id, **eng.last_timing})
                        del TIMING_LOG[:-4096]
                    out.extend(t if isinstance(t, Exception) else completion(n_ids[j], t) for j, t in zip(idx, outs))
            return out
        prepared = [f.result() for f in futs]
        toks_out: List[Optional[List[int]]] = [None] * len(prepared)
        with eng.lock:
            with_img = [i for i, (_, f) in enumerate(prepared) if f is not None]
            for g0 in range(0, len(with_img), eng.max_batch):
                grp = with_img[g0:g0 + eng.max_batch]
                reqs = [(prepared[i][0], _frame_to_device(prepared[i][1], eng.device)) for i in grp]
                outs = eng.generate_batch(reqs, max_new_tokens=max_new, temperature=temp, seed=self.seed,
                                          stop_on_eos=not ignore_eos)
                for i, t in zip(grp, outs):
                    toks_out[i] = t
            for i, (ids, f) in enumerate(prepared):
                if f is None:
                    toks_out[i] = eng.generate(ids, None, max_new_tokens=max_new, temperature=temp, seed=self.seed,
                                               stop_on_eos=not ignore_eos)
        return [completion(len(ids), t) for (ids, _), t in zip(prepared, toks_out)]


_MOCK_REPLY: List[Optional[Any]] = [None]


def set_mock_reply(reply) -> None:
    """Reply (string, or callable messages -> string) of every ``CannedResponseClient`` built without one - i.e.
    of the clients the agents build under provider ``mock`` (None restores the default "OK")."""
    _MOCK_REPLY[0] = reply


class CannedResponseClient:
    """Mock backend (the reference declares ``use_mock_responses``, utils/config.py:191, but ships none):
    returns a fixed reply.  Used for the no-GPU plumbing configuration (BASELINE config 1) and by tests;
    it performs no model arithmetic and is never selected implicitly."""

    def __init__(self, reply: Optional[str] = None, **_ignored):
        self.reply = reply if reply is not None else (_MOCK_REPLY[0] if _MOCK_REPLY[0] is not None else "OK")
        self.calls: List[dict] = []
        self.chat = _Chat(self)

    def _complete(self, model, messages, temperature, max_tokens, **kwargs) -> ChatCompletion:
        self.calls.append({"model": model, "messages": messages, "temperature": temperature, "max_tokens": max_tokens})
        reply = self.reply(messages) if callable(self.reply) else self.reply
        return ChatCompletion([_Choice(_Message(reply))], model=model or "")


def make_client(provider: str, api_key: Optional[str] = None, **kwargs):
    """provider -> client object.  ``mi355x`` -> local engine; ``mock`` -> canned replies;
    ``huggingface`` -> the reference's own remote client (only if huggingface_hub is importable)."""
    if provider == LOCAL_PROVIDER:
        return LocalVLMClient(api_key=api_key, **kwargs)
    if provider == "mock":
        return CannedResponseClient(**kwargs)
    if provider == "huggingface":
        from huggingface_hub import InferenceClient
        return InferenceClient(api_key=api_key)
    raise ValueError(f"unknown provider {provider!r}")
