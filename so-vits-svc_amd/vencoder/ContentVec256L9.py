"""Mirror of vencoder/ContentVec256L9.py: the `vec256l9` speech encoder — transformer layer 9 of the ContentVec model followed
by its `final_proj` Linear(768, 256) (:36-37) — on the MI355X engine; see ContentVec768L12.py."""
from vencoder.ContentVec768L12 import ContentVec768L12


class ContentVec256L9(ContentVec768L12):
    OUTPUT_LAYER = 9
    USE_FINAL_PROJ = True
