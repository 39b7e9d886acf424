"""Model families, selected by ``[config] model`` exactly as the reference's ``model.<name>`` packages
(train.py:96, detect.py:96).  Only yolo2 is on the MI355X hot path."""
