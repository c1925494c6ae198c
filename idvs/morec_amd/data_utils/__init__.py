from .preprocess import read_news, read_news_bert, get_doc_input_bert, read_behaviors, read_images  # noqa: F401
from .dataset import BuildTrainDataset, BuildEvalDataset, SequentialDistributedSampler, collate_train_batch, collate_bce_batch, epoch_batches  # noqa: F401
from .metrics import eval_model, get_item_embeddings  # noqa: F401
from .utils import get_checkpoint, load_model, save_model  # noqa: F401
from .images import DeviceImageFeed, LMDB_Image, LmdbImageStore, LmdbItemImages, PatchRows, decode_record, pack_images, resize_table  # noqa: F401
