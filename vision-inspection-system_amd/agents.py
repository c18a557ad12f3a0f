"""Boundary B2: the Inspector / Auditor agents with the reference's constructor, attributes, method
signatures and never-raise error behaviour, talking to whichever client the configured provider selects.

Mirrors src/agents/vlm_inspector.py (``VLMInspectorAgent``: ctor :30-44, ``_encode_image_optimized`` :46-88,
``_call_api_with_retry`` :90-140, ``analyze`` :433-526, ``health_check`` :528-554),
src/agents/vlm_auditor.py (``VLMAuditorAgent``: ctor/_init_client :29-83, ``verify`` :166-234,
``health_check`` :498-528), src/agents/base.py:25-47 (attributes) and src/agents/__init__.py:11-18 (factories).
With provider ``mi355x`` the client is ``LocalVLMClient`` (HIP engine); with ``huggingface`` it is the
reference's own remote client, so the same agent code serves both backends.
"""
from __future__ import annotations

import logging
import time
from pathlib import Path
from typing import Any, Dict, Optional

from .client import make_client
from .config import LOCAL_PROVIDER, get_config
from .image_processing import direct_frames_enabled, encode_image_optimized, frame_url_for, release_frames
from .prompts import AUDITOR_PROMPT, INSPECTOR_PROMPT
from .response_parsing import parse_json_robust, validate_and_fix_result
from .schemas import InspectionContext, VLMAnalysisResult


def _logger(name: str) -> logging.Logger:
    try:  # pragma: no cover - only inside the reference application
        from utils.logger import setup_logger  # type: ignore
        return setup_logger(f"agent.{name}", level=get_config().log_level, component=name.upper())
    except Exception:
        return logging.getLogger(f"agent.{name}")


class _BaseAgent:
    """Attribute contract of BaseVLMAgent (src/agents/base.py:25-47)."""

    def __init__(self, nickname: str):
        self.llm = None
        self.nickname = nickname
        self.is_vision = True
        self.logger = _logger(nickname)

    def _call_with_retry(self, messages: list, max_retries: int = 3, classify: bool = True) -> str:
        """3 attempts, 1-2-4 s backoff.  ``classify`` (Inspector): "429"/"rate" -> back off and retry,
        "413"/"payload" -> ValueError without retry (vlm_inspector.py:113-140); the Auditor retries every
        error alike (vlm_auditor.py:150-164)."""
        delay = 1
        for attempt in range(max_retries):
            try:
                completion = self.client.chat.completions.create(
                    model=self.model_id, messages=messages, temperature=self.temperature, max_tokens=self.max_tokens)
                return completion.choices[0].message.content
            except Exception as e:
                text = str(e)
                if classify and ("429" in text or "rate" in text.lower()):
                    wait = delay * (2 ** attempt)
                    self.logger.warning(f"Rate limited, waiting {wait}s before retry {attempt + 1}/{max_retries}")
                    time.sleep(wait)
                elif classify and ("413" in text or "payload" in text.lower()):
                    raise ValueError(f"Image payload too large for API: {e}")
                elif attempt < max_retries - 1:
                    wait = delay * (2 ** attempt)
                    self.logger.warning(f"API attempt {attempt + 1} failed: {e}, retrying in {wait}s...")
                    time.sleep(wait)
                else:
                    raise
        raise RuntimeError(f"API call failed after {max_retries} attempts")

    def health_check(self) -> bool:
        """Text-only ping, ``max_tokens=10``, no temperature (vlm_inspector.py:528-554)."""
        try:
            completion = self.client.chat.completions.create(
                model=self.model_id, messages=[{"role": "user", "content": "Respond with only the word 'OK'"}],
                max_tokens=10)
            content = completion.choices[0].message.content
            ok = bool(content and len(content) > 0)
            if ok:
                self.logger.info(f"{self.nickname} is healthy")
            return ok
        except Exception as e:
            self.logger.error(f"{self.nickname} health check failed: {e}")
            return False


class VLMInspectorAgent(_BaseAgent):
    """Primary inspection agent (no-argument constructor, reads the global config)."""

    def __init__(self):
        cfg = get_config()
        self.provider = getattr(cfg, "vlm_inspector_provider", "huggingface")
        self.client = make_client(self.provider, api_key=getattr(cfg, "huggingface_api_key", None))
        self.model_id = cfg.vlm_inspector_model
        self.temperature = cfg.vlm_inspector_temperature
        self.max_tokens = cfg.vlm_inspector_max_tokens
        self.max_image_size = getattr(cfg, "max_image_dimension", 1024)
        super().__init__("Inspector")
        self.logger.info(f"Initialized Inspector with {self.provider} model: {self.model_id}")

    def _encode_image_optimized(self, image_path: Path, max_size: Optional[int] = None) -> str:
        if self.provider == LOCAL_PROVIDER and direct_frames_enabled():      # opt-in: pixels handed over in-process
            return frame_url_for(image_path, max_size or self.max_image_size, convert_la=True, logger=self.logger)
        return encode_image_optimized(image_path, max_size or self.max_image_size, convert_la=True,
                                      enforce_limit=True, logger=self.logger)

    def _call_api_with_retry(self, messages: list, max_retries: int = 3) -> str:
        return self._call_with_retry(messages, max_retries, classify=True)

    def _parse_json_robust(self, text: str) -> Dict[str, Any]:
        return parse_json_robust(text, rescue_partial=True, logger=self.logger)

    def _validate_and_fix_result(self, result_dict: Dict[str, Any]) -> Dict[str, Any]:
        return validate_and_fix_result(result_dict, logger=self.logger)

    def _messages(self, image_path: Path, context: InspectionContext) -> list:
        prompt = INSPECTOR_PROMPT.format(criticality=context.criticality, domain=context.domain or "general",
                                         user_notes=context.user_notes or "None provided")
        image_data = self._encode_image_optimized(image_path)
        return [{"role": "user", "content": [{"type": "text", "text": prompt},
                                             {"type": "image_url", "image_url": {"url": image_data}}]}]

    def _interpret(self, response_text: str, context: InspectionContext) -> VLMAnalysisResult:
        result = VLMAnalysisResult(**self._validate_and_fix_result(self._parse_json_robust(response_text)))
        if result.inferred_criticality and result.inferred_criticality != context.criticality:
            self.logger.info(f"Agent inferred criticality '{result.inferred_criticality}' differs from "
                             f"user's '{context.criticality}'")
        self.logger.info(f"Analysis complete: {len(result.defects)} defects found, "
                         f"confidence: {result.overall_confidence}")
        return result

    @staticmethod
    def _failure(e: Exception) -> VLMAnalysisResult:
        return VLMAnalysisResult(
            object_identified="unknown", overall_condition="uncertain", defects=[], overall_confidence="low",
            analysis_reasoning=f"Analysis failed due to error: {str(e)}", analysis_failed=True,
            failure_reason=f"Inspector analysis failed: {str(e)}")

    def analyze(self, image_path: Path, context: InspectionContext) -> VLMAnalysisResult:
        """Image + context -> structured result.  Never raises: any failure becomes an
        ``analysis_failed=True`` result (vlm_inspector.py:515-526)."""
        self.logger.info(f"Starting inspection for image: {context.image_id}")
        messages = None
        try:
            messages = self._messages(image_path, context)
            t0 = time.time()
            response_text = self._call_api_with_retry(messages)
            self.logger.info(f"{self.provider} response received in {time.time() - t0:.2f}s")
            return self._interpret(response_text, context)
        except Exception as e:
            self.logger.error(f"Inspector analysis failed: {e}", exc_info=True)
            return self._failure(e)
        finally:
            release_frames(messages)       # VIS_DIRECT_FRAMES handles end with the request (all retries included)

    def analyze_many(self, image_paths, contexts, prepared=None) -> list:
        """Batch form of ``analyze`` for clients that can serve several requests with one shared decode loop
        (LocalVLMClient.complete_many).  Failures stay per image; never raises.  ``prepared``: futures from
        ``prepare_many`` (requests encoded ahead of time on the ingest pool)."""
        return _many(self, image_paths, contexts, prepared)

    def prepare_many(self, image_paths, contexts) -> list:
        """Submit the request-side work of every image (open, thumbnail, JPEG q85, base64, prompt) to the ingest pool;
        returns one future per image for ``analyze_many(..., prepared=)``."""
        from . import ingest
        return [ingest.submit(_traced_messages, self, Path(p), c) for p, c in zip(image_paths, contexts)]


class VLMAuditorAgent(_BaseAgent):
    """Independent second opinion; same contract, own prompt and decoding settings."""

    def __init__(self):
        cfg = get_config()
        self.model_id = cfg.vlm_auditor_model
        self.temperature = cfg.vlm_auditor_temperature
        self.max_tokens = cfg.vlm_auditor_max_tokens
        self.provider = getattr(cfg, "vlm_auditor_provider", "groq")
        self.use_groq = False
        self.use_huggingface = False
        super().__init__("Auditor")
        self._init_client(cfg)
        self.logger.info(f"Initialized Auditor with {self.provider.upper()} model: {self.model_id}")

    def _init_client(self, cfg) -> None:
        if self.provider in (LOCAL_PROVIDER, "mock"):
            self.client = make_client(self.provider)
            return
        if self.provider == "groq":
            try:
                from groq import Groq  # type: ignore
                self.client = Groq(api_key=getattr(cfg, "groq_api_key", None))
                self.use_groq = True
                return
            except Exception as e:
                self.logger.warning(f"Groq SDK unavailable ({e}), falling back to HuggingFace")
        self.client = make_client("huggingface", api_key=getattr(cfg, "huggingface_api_key", None))
        self.use_huggingface = True
        if "llama-4" in self.model_id.lower():
            self.model_id = "meta-llama/Llama-3.2-11B-Vision-Instruct"

    def _encode_image_optimized(self, image_path: Path, max_size: int = 1024) -> str:
        if self.provider == LOCAL_PROVIDER and direct_frames_enabled():
            return frame_url_for(image_path, max_size, convert_la=False, logger=self.logger)
        return encode_image_optimized(image_path, max_size, convert_la=False, enforce_limit=False, logger=self.logger)

    def _parse_json_robust(self, text: str) -> Dict[str, Any]:
        return parse_json_robust(text, rescue_partial=False, logger=self.logger)

    def _validate_and_fix_result(self, result_dict: Dict[str, Any]) -> Dict[str, Any]:
        return validate_and_fix_result(result_dict, logger=self.logger, who="auditor ")

    def _messages(self, image_path: Path, context: InspectionContext) -> list:
        prompt = AUDITOR_PROMPT.format(criticality=context.criticality, domain=context.domain or "general")
        image_data = self._encode_image_optimized(image_path)
        return [{"role": "user", "content": [{"type": "text", "text": prompt},
                                             {"type": "image_url", "image_url": {"url": image_data}}]}]

    def _interpret(self, response_text: str, context: InspectionContext) -> VLMAnalysisResult:
        result = VLMAnalysisResult(**self._validate_and_fix_result(self._parse_json_robust(response_text)))
        self.logger.info(f"Audit complete: {len(result.defects)} defects found, "
                         f"confidence: {result.overall_confidence}")
        return result

    @staticmethod
    def _failure(e: Exception) -> VLMAnalysisResult:
        return VLMAnalysisResult(
            object_identified="unknown", overall_condition="uncertain", defects=[], overall_confidence="low",
            analysis_reasoning=f"Audit verification failed: {str(e)}", analysis_failed=True,
            failure_reason=f"Auditor verification failed: {str(e)}")

    def verify(self, image_path: Path, context: InspectionContext,
               inspector_result: VLMAnalysisResult) -> VLMAnalysisResult:
        """Independent analysis of the same image; ``inspector_result`` is accepted and - as in the
        reference (vlm_auditor.py:187-191) - not shown to the model.  Never raises."""
        self.logger.info(f"Starting audit verification for: {context.image_id}")
        messages = None
        try:
            messages = self._messages(image_path, context)
            t0 = time.time()
            response_text = self._call_with_retry(messages, 3, classify=False)
            self.logger.info(f"Auditor response received in {time.time() - t0:.2f}s")
            return self._interpret(response_text, context)
        except Exception as e:
            self.logger.error(f"Auditor verification failed: {e}", exc_info=True)
            return self._failure(e)
        finally:
            release_frames(messages)

    def verify_many(self, image_paths, contexts, inspector_results=None, prepared=None) -> list:
        """Batch form of ``verify`` (see VLMInspectorAgent.analyze_many)."""
        return _many(self, image_paths, contexts, prepared)

    def prepare_many(self, image_paths, contexts) -> list:
        from . import ingest
        return [ingest.submit(_traced_messages, self, Path(p), c) for p, c in zip(image_paths, contexts)]


def _traced_messages(agent, path, context):
    from . import ingest
    with ingest.span("request-side a3 encode (PIL thumbnail / JPEG q85 / base64, pool thread)"):
        return agent._messages(path, context)


def _many(agent, image_paths, contexts, prepared=None) -> list:
    """Shared body of analyze_many / verify_many: encode every image on the ingest pool (failures stay per image), one
    ``complete_many`` call when the client offers it (one shared decode loop), per-reply interpretation.
    A client that sets ``accepts_futures`` (LocalVLMClient) is handed the encode futures themselves and starts its
    first prompt pass while the later images are still being encoded; it returns the exception in the place of a
    request whose encode or decode failed."""
    from . import ingest
    results = [None] * len(image_paths)
    futs = prepared if prepared is not None else \
        [ingest.submit(_traced_messages, agent, Path(p), c) for p, c in zip(image_paths, contexts)]
    streaming = getattr(agent.client, "accepts_futures", False) and hasattr(agent.client, "complete_many")
    todo, msgs = [], []
    for i, (path, fut) in enumerate(zip(image_paths, futs)):
        if streaming:
            msgs.append(fut)
            todo.append(i)
            continue
        ok, val = ingest.outcome(fut)
        if ok:
            msgs.append(val)
            todo.append(i)
        else:
            agent.logger.error(f"{agent.nickname}: request for {path} failed: {val}")
            results[i] = agent._failure(val)
    if todo:
        try:
            if hasattr(agent.client, "complete_many"):
                replies = agent.client.complete_many(agent.model_id, msgs, agent.temperature, agent.max_tokens)
                texts = [r if isinstance(r, Exception) else r.choices[0].message.content for r in replies]
            else:
                texts = [agent.client.chat.completions.create(model=agent.model_id, messages=m,
                                                              temperature=agent.temperature,
                                                              max_tokens=agent.max_tokens).choices[0].message.content
                         for m in msgs]
        except Exception as e:
            agent.logger.error(f"{agent.nickname}: batched call failed: {e}", exc_info=True)
            for i in todo:
                results[i] = agent._failure(e)
            return results
        for i, text in zip(todo, texts):
            if isinstance(text, Exception):
                agent.logger.error(f"{agent.nickname}: request for {image_paths[i]} failed: {text}")
                results[i] = agent._failure(text)
                continue
            try:
                with ingest.span("calling thread: parse + validate a reply"):
                    results[i] = agent._interpret(text, contexts[i])
            except Exception as e:
                results[i] = agent._failure(e)
    for fut in futs:        # the batch call is over: VIS_DIRECT_FRAMES handles of its requests are released
        if fut.done() and not fut.cancelled() and fut.exception() is None:
            release_frames(fut.result())
    return results


InspectorAgent = VLMInspectorAgent
AuditorAgent = VLMAuditorAgent


def get_inspector() -> VLMInspectorAgent:
    """New agent object per call, like the reference (src/agents/__init__.py:11-13); the model itself
    lives in the process-wide engine registry (client.get_model)."""
    return VLMInspectorAgent()


def get_auditor() -> VLMAuditorAgent:
    return VLMAuditorAgent()


def health_check_agents() -> dict:
    """name -> (ok, details) for the two vision agents (src/agents/__init__.py:26-68, Explainer out of scope)."""
    cfg = get_config()
    results = {}
    for label, factory, model in (("Inspector", get_inspector, cfg.vlm_inspector_model),
                                  ("Auditor", get_auditor, cfg.vlm_auditor_model)):
        try:
            ok = factory().health_check()
            results[label] = (ok, f"Model: {model}" if ok else "Connection failed")
        except Exception as e:
            results[label] = (False, f"Error: {e}")
    return results
