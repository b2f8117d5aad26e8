"""stdlib logging, same logger name as the reference (bfcnn/custom_logger.py:8-13)."""
import logging

logging.basicConfig(level=logging.WARNING,
                    format="%(asctime)s %(levelname)-8s %(message)s")
logger = logging.getLogger("bfcnn")
