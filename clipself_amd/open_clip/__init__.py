from .factory import (create_model, create_model_and_transforms, get_cast_dtype, get_tokenizer, list_models,
                      load_checkpoint, load_state_dict)
from .model import CLIP, ClipVisionTower, CustomCLIP, EVAVisionTower, OPENAI_DATASET_MEAN, OPENAI_DATASET_STD, boxes_to_rois
