"""Package logger (the reference configures a root logger at DEBUG, Skps/logger/logger.py:11-25;
here a named logger at INFO so importing the package does not hijack the host application's logging)."""
import logging

logger = logging.getLogger("peppa_hip")
if not logger.handlers:
    _h = logging.StreamHandler()
    _h.setFormatter(logging.Formatter("%(asctime)s %(name)s %(levelname)s: %(message)s"))
    logger.addHandler(_h)
logger.setLevel(logging.WARNING)
