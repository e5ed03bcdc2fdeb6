"""Checkpoint I/O in the reference's `.pkl` format (helen/modules/python/models/ModelHander.py:38-133):
a torch.save'd dict {model_state_dict, model_optimizer, hidden_size, gru_layers, epochs}."""
import os
from collections import OrderedDict

import numpy as np
import torch

from .transducer import TransducerGRU


class ModelHandler(object):
    @staticmethod
    def get_new_gru_model(input_channels, image_features, gru_layers, hidden_size, num_base_classes,
                          num_rle_classes):
        return TransducerGRU(input_channels, image_features, gru_layers, hidden_size,
                             num_base_classes, num_rle_classes, bidirectional=True)

    @staticmethod
    def load_simple_model(model_path, input_channels, image_features, seq_len, num_base_classes,
                          num_rle_classes):
        """-> (model, hidden_size, gru_layers, epochs), like ModelHander.py:38-82: reads the dict
        on the host, honours its hidden_size / gru_layers, strips a leading `module.` left by
        DataParallel/DDP training and loads the state into a TransducerGRU."""
        try:
            checkpoint = torch.load(model_path, map_location="cpu", weights_only=True)
        except Exception:
            checkpoint = torch.load(model_path, map_location="cpu", weights_only=False)
        hidden_size = checkpoint["hidden_size"]
        gru_layers = checkpoint["gru_layers"]
        epochs = checkpoint["epochs"]
        model = ModelHandler.get_new_gru_model(input_channels, image_features, gru_layers,
                                               hidden_size, num_base_classes, num_rle_classes)
        state = OrderedDict()
        for k, v in checkpoint["model_state_dict"].items():
            state[k[7:] if k[0:7] == "module." else k] = v
        model.load_state_dict(state)
        return model, hidden_size, gru_layers, epochs

    @staticmethod
    def save_model(transducer_model, model_optimizer, hidden_size, layers, epoch, file_name):
        """Write a checkpoint the reference's loader accepts (ModelHander.py:109-133)."""
        if os.path.isfile(file_name):
            os.remove(file_name)
        sd = transducer_model.state_dict() if hasattr(transducer_model, "state_dict") \
            else transducer_model
        sd = OrderedDict((k, torch.from_numpy(np.array(v, dtype=np.float32))
                          if not isinstance(v, torch.Tensor) else v) for k, v in sd.items())
        opt = model_optimizer.state_dict() if hasattr(model_optimizer, "state_dict") \
            else (model_optimizer or {})
        torch.save({"model_state_dict": sd, "model_optimizer": opt, "hidden_size": hidden_size,
                    "gru_layers": layers, "epochs": epoch}, file_name)
