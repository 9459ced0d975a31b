"""Text retriever seam.

In the reference this component prompts a 7B vision-language model and WordNet
(/root/reference/mars/components/TextRetrieverModule.py:42-137); that stage lies before
``time_start_ranking_after_text_extraction`` (/root/reference/mars/MARS.py:59) and outside the
ranking path built here.  The interface is kept -- ``get_conceptual_information(support_images,
support_masks) -> (name, description)`` -- and this implementation returns the class string it was
configured with, which is how the synthetic benchmark and any caller with known class names drive
the path.  A VLM-backed retriever with the same method can be passed to ``MARS`` instead.
"""
from __future__ import annotations

from typing import Tuple


class TextRetrieverModule:
    def __init__(self, name: str = "object", description: str = ""):
        self.name = name
        self.description = description

    def set_class(self, name: str, description: str = ""):
        self.name, self.description = name, description

    def get_conceptual_information(self, support_images=None, support_masks=None) -> Tuple[str, str]:
        return self.name, self.description


def build_text_retriever_component(args):
    return TextRetrieverModule(getattr(args, "class_name", "object"), getattr(args, "class_description", ""))
