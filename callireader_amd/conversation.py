"""Prompt template used by the reference chat API: 'internlm2-chat' (MPT separator style).

Behaviour of /root/reference/InternVL/conversation.py:238-247 (get_prompt, MPT branch) and :358-374 (template
registration) restated for the one template the hot path uses.
"""
from dataclasses import dataclass, field
from typing import List, Optional


@dataclass
class Conversation:
    name: str = 'internlm2-chat'
    system_template: str = '<|im_start|>system\n{system_message}'
    system_message: str = '你是由上海人工智能实验室联合商汤科技开发的书生多模态大模型，英文名叫InternVL, 是一个有用无害的人工智能助手。'
    roles: tuple = ('<|im_start|>user\n', '<|im_start|>assistant\n')
    sep: str = '<|im_end|>'
    messages: List[List[Optional[str]]] = field(default_factory=list)

    def append_message(self, role, message):
        self.messages.append([role, message])

    def get_prompt(self):
        ret = self.system_template.format(system_message=self.system_message) + self.sep
        for role, message in self.messages:
            ret += (role + message + self.sep) if message else role
        return ret


def get_conv_template(name):
    if name != 'internlm2-chat':
        raise KeyError(f'only the internlm2-chat template is on the CalliReader path, got {name!r}')
    return Conversation()
