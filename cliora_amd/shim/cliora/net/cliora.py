"""`from cliora.net.cliora import DioraMLP as Diora` (cliora/net/trainer.py:521) -> the native vision-language chart module."""
from cliora_amd.cliora import AttentionHead, DioraMLP, VLComposeMLP  # noqa: F401
