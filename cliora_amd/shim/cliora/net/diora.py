"""`from cliora.net.diora import DioraMLP as Diora` (cliora/net/trainer.py:523) -> the native text-only chart module."""
from cliora_amd.diora import Bilinear, Chart, ComposeMLP, DioraBase, DioraMLP  # noqa: F401
from cliora_amd.treelstm import DioraTreeLSTM  # noqa: F401
