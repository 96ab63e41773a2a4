"""Selection shim (SURVEY.md section 8b "Selection"): put this directory BEFORE the reference checkout on sys.path and
`cliora.net.diora` / `cliora.net.cliora` resolve to the MI355X-native modules while every other `cliora.*` module still
comes from the reference -- `cliora/net/trainer.py:518-526` (build_net) then needs no edit."""
from pkgutil import extend_path

__path__ = extend_path(__path__, __name__)
