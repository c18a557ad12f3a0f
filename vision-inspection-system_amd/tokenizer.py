"""Text <-> token ids and the Qwen2-VL chat layout.

Two tokenizers behind one interface:
  * ``HFTokenizer`` wraps a LOCAL ``tokenizer.json`` through the ``tokenizers`` library (no hub access);
    this is what a real Qwen2-VL model directory uses.
  * ``ByteTokenizer`` is a self-contained byte-level vocabulary (256 bytes + specials) for the tiny
    synthetic model and the benchmark, where no vocabulary file exists offline.

``build_chat_ids`` lays a ``messages`` list out the way the Qwen2-VL chat template does
(``<|im_start|>role\\n ... <|im_end|>\\n``, images as ``<|vision_start|><|image_pad|>*n<|vision_end|>`` in content
order, default system prompt when the conversation has none, generation prompt ``<|im_start|>assistant\\n``).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence, Tuple

DEFAULT_SYSTEM = "You are a helpful assistant."


class ByteTokenizer:
    """ids 0..255 = raw UTF-8 bytes; specials sit above 255.  Needs vocab >= 264."""

    def __init__(self, vocab: int, image_token_id: int, vision_start_id: int, vision_end_id: int,
                 eos_ids: Sequence[int]):
        if vocab < 264:
            raise ValueError("ByteTokenizer needs a vocabulary of at least 264 ids")
        self.vocab = vocab
        self.im_start_id = 256
        self.im_end_id = eos_ids[0]
        self.image_token_id = image_token_id
        self.vision_start_id = vision_start_id
        self.vision_end_id = vision_end_id
        self.eos_ids = tuple(eos_ids)
        self._special = {self.im_start_id, self.im_end_id, image_token_id, vision_start_id, vision_end_id, *eos_ids}

    def encode(self, text: str) -> List[int]:
        return list(text.encode("utf-8"))

    def decode(self, ids: Sequence[int]) -> str:
        return bytes(i for i in ids if 0 <= i < 256).decode("utf-8", errors="replace")


class HFTokenizer:
    def __init__(self, model_dir: str, image_token_id: int, vision_start_id: int, vision_end_id: int,
                 eos_ids: Sequence[int]):
        from tokenizers import Tokenizer
        path = os.path.join(model_dir, "tokenizer.json")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found (the local backend never downloads tokenizers)")
        self._tok = Tokenizer.from_file(path)
        self.vocab = self._tok.get_vocab_size()
        self.image_token_id = image_token_id
        self.vision_start_id = vision_start_id
        self.vision_end_id = vision_end_id
        self.eos_ids = tuple(eos_ids)
        self.im_start_id = self._tok.token_to_id("<|im_start|>")
        self.im_end_id = self._tok.token_to_id("<|im_end|>")
        if self.im_start_id is None or self.im_end_id is None:
            raise ValueError("tokenizer.json lacks <|im_start|>/<|im_end|> (not a Qwen2 chat vocabulary)")

    def encode(self, text: str) -> List[int]:
        return self._tok.encode(text, add_special_tokens=False).ids

    def decode(self, ids: Sequence[int]) -> str:
        return self._tok.decode(list(ids), skip_special_tokens=True)


def build_chat_ids(tok, messages: List[dict], image_token_counts: Sequence[int],
                   add_generation_prompt: bool = True) -> List[int]:
    """messages (OpenAI chat format) -> token ids; the i-th image gets image_token_counts[i] pad tokens."""
    ids: List[int] = []
    img_i = 0

    def role_block(role: str, body: List[int]):
        ids.append(tok.im_start_id)
        ids.extend(tok.encode(role + "\n"))
        ids.extend(body)
        ids.append(tok.im_end_id)
        ids.extend(tok.encode("\n"))

    if not messages or messages[0].get("role") != "system":
        role_block("system", tok.encode(DEFAULT_SYSTEM))
    for m in messages:
        body: List[int] = []
        content = m.get("content")
        if isinstance(content, str):
            body.extend(tok.encode(content))
        elif isinstance(content, list):
            for part in content:
                kind = part.get("type")
                if kind == "text":
                    body.extend(tok.encode(part.get("text", "")))
                elif kind in ("image_url", "image"):
                    if img_i >= len(image_token_counts):
                        raise ValueError("more image parts than decoded images")
                    body.append(tok.vision_start_id)
                    body.extend([tok.image_token_id] * int(image_token_counts[img_i]))
                    body.append(tok.vision_end_id)
                    img_i += 1
                else:
                    raise ValueError(f"unsupported content part type: {kind!r}")
        elif content is not None:
            raise ValueError("message content must be a string or a list of parts")
        role_block(m.get("role", "user"), body)
    if img_i != len(image_token_counts):
        raise ValueError("fewer image parts than decoded images")
    if add_generation_prompt:
        ids.append(tok.im_start_id)
        ids.extend(tok.encode("assistant\n"))
    return ids


# ----------------------------------------------------------------------------- Llama-3.2-Vision chat layout (row f2)
class LlamaByteTokenizer:
    """Byte-level stand-in for the Llama 3 vocabulary (tiny synthetic mllama model): ids 0..255 = raw bytes,
    chat specials above."""

    def __init__(self, vocab: int, image_token_id: int, eos_ids: Sequence[int]):
        if vocab < 264:
            raise ValueError("LlamaByteTokenizer needs a vocabulary of at least 264 ids")
        self.vocab = vocab
        self.bos_id, self.start_header_id, self.end_header_id, self.eot_id = 256, 257, 258, 259
        self.image_token_id = image_token_id
        self.eos_ids = tuple(eos_ids) + (self.eot_id,)

    def encode(self, text: str) -> List[int]:
        return list(text.encode("utf-8"))

    def decode(self, ids: Sequence[int]) -> str:
        return bytes(i for i in ids if 0 <= i < 256).decode("utf-8", errors="replace")


class LlamaHFTokenizer:
    def __init__(self, model_dir: str, image_token_id: int, eos_ids: Sequence[int]):
        from tokenizers import Tokenizer
        path = os.path.join(model_dir, "tokenizer.json")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} not found (the local backend never downloads tokenizers)")
        self._tok = Tokenizer.from_file(path)
        self.vocab = self._tok.get_vocab_size()
        self.image_token_id = image_token_id
        self.eos_ids = tuple(eos_ids)
        ids = {n: self._tok.token_to_id(n) for n in ("<|begin_of_text|>", "<|start_header_id|>", "<|end_header_id|>",
                                                     "<|eot_id|>")}
        if any(v is None for v in ids.values()):
            raise ValueError("tokenizer.json lacks the Llama 3 chat specials")
        self.bos_id, self.start_header_id = ids["<|begin_of_text|>"], ids["<|start_header_id|>"]
        self.end_header_id, self.eot_id = ids["<|end_header_id|>"], ids["<|eot_id|>"]

    def encode(self, text: str) -> List[int]:
        return self._tok.encode(text, add_special_tokens=False).ids

    def decode(self, ids: Sequence[int]) -> str:
        return self._tok.decode(list(ids), skip_special_tokens=True)


def build_llama_chat_ids(tok, messages: List[dict], n_images: int, add_generation_prompt: bool = True) -> List[int]:
    """messages -> ids in the Llama-3.2-Vision chat layout:
    ``<|begin_of_text|>`` then per message ``<|start_header_id|>role<|end_header_id|>\n\n`` + content parts in the
    order given (text trimmed, every image ONE ``<|image|>`` token) + ``<|eot_id|>``, generation prompt
    ``<|start_header_id|>assistant<|end_header_id|>\n\n``.  The reference sends [text, image] (vlm_auditor.py:121-126),
    so the image token follows the prompt text."""
    ids: List[int] = [tok.bos_id]
    seen = 0
    for m in messages:
        ids.append(tok.start_header_id)
        ids.extend(tok.encode(m.get("role", "user")))
        ids.append(tok.end_header_id)
        ids.extend(tok.encode("\n\n"))
        content = m.get("content")
        if isinstance(content, str):
            ids.extend(tok.encode(content.strip()))
        elif isinstance(content, list):
            for part in content:
                kind = part.get("type")
                if kind == "text":
                    ids.extend(tok.encode(part.get("text", "").strip()))
                elif kind in ("image_url", "image"):
                    ids.append(tok.image_token_id)
                    seen += 1
                else:
                    raise ValueError(f"unsupported content part type: {kind!r}")
        elif content is not None:
            raise ValueError("message content must be a string or a list of parts")
        ids.append(tok.eot_id)
    if seen != n_images:
        raise ValueError("image parts and decoded images do not match")
    if add_generation_prompt:
        ids.append(tok.start_header_id)
        ids.extend(tok.encode("assistant"))
        ids.append(tok.end_header_id)
        ids.extend(tok.encode("\n\n"))
    return ids
