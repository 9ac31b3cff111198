"""Oracle: OrderFormer's model (TEST INFRASTRUCTURE ONLY -- the product runs csrc/orderformer.hip).

Restates models/model.py:206-233: `Transformer` = nn.Linear(input_dim, model_dim) embedding -> nn.TransformerEncoder
(num_layers copies of nn.TransformerEncoderLayer(d_model, nhead, batch_first=True): post-norm, dim_feedforward 2048,
ReLU, LayerNorm eps 1e-5; dropout is inactive in eval) -> nn.Linear(model_dim, output_dim) decoder; the reference
builds it with norm=False (no final LayerNorm, :530-552) in bf16 (it is constructed inside a bf16 from_pretrained)
and feeds it bf16 inputs (:458).  Built from the same torch modules, so on the CPU it is bit-identical to the reference
(pinned by tests/golden/ordering_vectors.json `_model`).
"""
import torch
import torch.nn as nn


class OrderFormerModel(nn.Module):
    def __init__(self, input_dim=4, model_dim=256, num_heads=8, num_layers=4, output_dim=1):
        super().__init__()
        self.embedding = nn.Linear(input_dim, model_dim)                                              # :210
        layer = nn.TransformerEncoderLayer(d_model=model_dim, nhead=num_heads, batch_first=True)      # :213
        self.transformer_encoder = nn.TransformerEncoder(layer, num_layers=num_layers, norm=None)     # :214, norm=False
        self.decoder = nn.Linear(model_dim, output_dim)                                               # :216

    def forward(self, x):                                                                             # :218-222
        return self.decoder(self.transformer_encoder(self.embedding(x)))


def build(state_dict, dtype=torch.bfloat16):
    m = OrderFormerModel().to(dtype)
    own = {k: v for k, v in state_dict.items() if not k.startswith('encoder_layer.')}     # the template layer never runs
    m.load_state_dict(own)
    return m.eval()


class CpuScorer:
    """Stands in for `Engine` in host-logic tests: `.orderformer(x)` runs the oracle model on the CPU."""

    def __init__(self, state_dict):
        self.model = build(state_dict)

    @torch.no_grad()
    def orderformer(self, x):
        return self.model(x.to(torch.bfloat16)).float().reshape(x.shape[0], x.shape[1])
