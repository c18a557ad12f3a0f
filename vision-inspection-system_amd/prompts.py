"""Prompt text for the two vision agents (row a4).

In a drop-in deployment the host application's own ``utils.prompts.INSPECTOR_PROMPT`` /
``AUDITOR_PROMPT`` (utils/prompts.py:18-95,:101-174) are used verbatim - the prompt is data the
application owns.  Stand-alone (tests, benchmark) the compact built-in prompts below are used; they
take the same ``str.format`` fields ({criticality}, {domain}, {user_notes} / {criticality}, {domain})
and ask for the same JSON schema, but are written independently.
"""

try:  # pragma: no cover - only inside the reference application
    from utils.prompts import INSPECTOR_PROMPT, AUDITOR_PROMPT  # type: ignore
    HOST_PROMPTS = True
except Exception:
    HOST_PROMPTS = False

    _SCHEMA = """{{
  "object_identified": "<what the part is>",
  "overall_condition": "damaged" | "good" | "uncertain",
  "defects": [
    {{
      "type": "<defect kind>",
      "location": "<where on the part>",
      "bbox": {{"x": <0-100>, "y": <0-100>, "width": <0-100>, "height": <0-100>}},
      "safety_impact": "CRITICAL" | "MODERATE" | "COSMETIC",
      "reasoning": "<one or two sentences>",
      "confidence": "high" | "medium" | "low",
      "recommended_action": "<what to do>"
    }}
  ],
  "overall_confidence": "high" | "medium" | "low",
  "analysis_reasoning": "<two or three sentence summary>"
}}"""

    INSPECTOR_PROMPT = (
        "You inspect a photographed part for defects on behalf of a safety team.\n"
        "Inspection context - criticality: {criticality}; domain: {domain}; operator notes: {user_notes}.\n\n"
        "Name the part, then look over the whole frame and list every defect you can actually see "
        "(structural damage, surface damage, wear, corrosion, contamination, missing or misassembled pieces). "
        "Do not invent defects; seams, reflections and shadows are not defects. For each defect give its kind, "
        "where it is, a tight bounding box, its safety impact (CRITICAL = could cause injury or failure, "
        "MODERATE = affects function or durability, COSMETIC = appearance only), a short reason, your confidence "
        "and a recommended action. Bounding boxes are PERCENTAGES of the image (0-100, origin top-left, "
        "x + width <= 100, y + height <= 100), never pixels. If the part looks sound and the image is clear, "
        "return an empty defect list with overall_condition \"good\" and high confidence.\n\n"
        "Answer with JSON only, in exactly this shape:\n" + _SCHEMA)

    AUDITOR_PROMPT = (
        "You are the second, independent reviewer of a photographed part; you have not seen any earlier findings.\n"
        "Inspection context - criticality: {criticality}; domain: {domain}.\n\n"
        "Identify the part and report only defects that are clearly visible, each with kind, location, a tight "
        "bounding box in PERCENTAGES of the image (0-100, origin top-left), safety impact (CRITICAL / MODERATE / "
        "COSMETIC), a short reason, confidence (high / medium / low) and a recommended action. Be sceptical: "
        "normal manufacturing features, glare and shadows are not defects. A clean part is reported with an empty "
        "defect list, overall_condition \"good\" and high confidence.\n\n"
        "Answer with JSON only, in exactly this shape:\n" + _SCHEMA)

__all__ = ["INSPECTOR_PROMPT", "AUDITOR_PROMPT", "HOST_PROMPTS"]
