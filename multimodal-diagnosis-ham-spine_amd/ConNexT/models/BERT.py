"""BERT CLS encoder of the ConNeXT model (reference ConNexT/models/BERT.py:7-22)."""
import torch.nn as nn

from hamspine import small as S
from hamspine.nn import BertModel


class BertEncoder(nn.Module):
    def __init__(self, model_path="/data/QLI/BERT_pretain"):
        super().__init__()
        self.bert = BertModel.from_pretrained(model_path)

    def forward(self, input_ids, attention_mask=None):
        hidden = self.bert(input_ids=input_ids, attention_mask=attention_mask).last_hidden_state
        return S.select_token(hidden, 0)   # (B, hidden) f32
