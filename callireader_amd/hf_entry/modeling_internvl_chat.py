"""Drop-in for <checkpoint>/modeling_internvl_chat.py: the file the checkpoint's config.json names in `auto_map`
(/root/reference/InternVL/config.json:6-10: "AutoModel": "modeling_internvl_chat.InternVLChatModel").

Copy this file (and configuration shims are not needed: config.json's own `AutoConfig` entry keeps working) over the
checkpoint directory's modeling_internvl_chat.py and the reference's unchanged call

    model = AutoModel.from_pretrained(INTERNVL_PATH, torch_dtype=torch.bfloat16, low_cpu_mem_usage=True,
                                      trust_remote_code=True).eval().cuda()          # inference.py:85-89

returns the MI355X engine's InternVLChatModel.  `callireader_amd` must be importable (repo root on PYTHONPATH).
"""
from callireader_amd.modeling_internvl_chat import InternVLChatModel  # noqa: F401
