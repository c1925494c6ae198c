from .model import Model  # noqa: F401  (same import surface as the reference: ``from model import Model``)
from .encoders import Bert_Encoder, IdEmbedding, Text_Encoder, User_Encoder  # noqa: F401
from .bert import HipBertModel  # noqa: F401
from .spec import BertShape  # noqa: F401
from .bce import BceModel  # noqa: F401  (drop-in for bce_text/main-end2end/model/model.py)
