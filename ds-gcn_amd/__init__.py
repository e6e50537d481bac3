"""dsgcn_amd — MI355X-native DS-GCN hot path behind the PYSKL registry/config API.

Drop-in names (reference: pyskl/models/builder.py, pyskl/models/gcns/*, recognizers/*, heads/*,
losses/*): ``build_model``, ``RecognizerGCN``, ``DGSTGCN``, ``STGCN``, ``CTRGCN``, ``GCNHead``,
``CrossEntropyLoss``, ``Graph``, ``Config``.
"""
from .registry import Registry, build_from_cfg
from .config import Config, ConfigDict
from .graph import Graph
from .builder import (MODELS, BACKBONES, HEADS, LOSSES, NECKS, RECOGNIZERS, build_backbone, build_head, build_loss,
                      build_model, build_recognizer)
from .evaluation import top_k_accuracy, mean_class_accuracy, confusion_matrix
from .losses import CrossEntropyLoss
from .heads import GCNHead, SimpleHead
from .gcn_units import dggcn, dgphgcn1, unit_aagcn, unit_gcn, unit_ctrgcn, unit_ctrhgcn, CTRGC, CTRHGC, Deferred
from .tcn_units import dgmstcn, mstcn, msmlp, unitmlp, unit_tcn, MSTCN
from .backbones import DGSTGCN, STGCN, CTRGCN, AAGCN, DGBlock, STGCNBlock, CTRGCNBlock, AAGCNBlock
from .recognizers import RecognizerGCN, reduce_log_vars, gather_results
from . import kernels
from . import pipeline
from .pipeline import PIPELINES, DATASETS, Compose, PoseDataset, SkeletonStore, SkeletonBatcher, build_dataset
from .data_parallel import FlatParams, FlatDataParallel, shard_batch
from .train import FlatSGD, cosine_lr
from .checkpoint import load_checkpoint, save_checkpoint, resume, find_resume, fuse_conv_bn
from .engine import TrainEngine
from .apis import train_model, EpochRunner, epoch_indices

__version__ = '0.1.0'
