"""Drop-in for <checkpoint>/tokenization_internlm2.py (tokenizer_config.json auto_map: "AutoTokenizer":
["tokenization_internlm2.InternLM2Tokenizer", null]): `AutoTokenizer.from_pretrained(INTERNVL_PATH,
trust_remote_code=True)` (inference.py:90) then returns the engine's sentencepiece-free reader of tokenizer.model."""
from callireader_amd.tokenization_internlm2 import InternLM2Tokenizer  # noqa: F401
