"""OK-VQA answer post-processing (ref:icv_src/metrics/okvqa_utils.py:205-215, ref:utils.py:128-133): cut the generation at the
next prompt keyword / first ", ", then stem every word the way the OK-VQA annotations were stemmed.

The stemmer needs three things this image does not have (no network): `nltk` with its tokenizer, POS-tagger and WordNet data,
and `inflection`.  They are imported when the first answer is stemmed; if any is missing the call raises an ImportError that
says which — it never falls back to an unstemmed answer, which would silently lower every OK-VQA score.  Parity of this module
is therefore unpinned here: the rule order below follows the reference text, the per-word
exception table is read from the file named by LICV_OKVQA_MANUAL_MATCHES (JSON {"word": "stem", ...}) when set."""
from __future__ import annotations

import json
import os
import re
from typing import Dict, Optional


class OKVQAStemmer:
    def __init__(self, manual_matches: Optional[Dict[str, str]] = None):
        self._manual = manual_matches
        self._nltk = self._inflection = self._lemmatizer = None

    def _load(self):
        if self._nltk is not None:
            return
        try:
            import nltk
            from nltk.stem import WordNetLemmatizer
        except ImportError as e:                                              # pragma: no cover - depends on the image
            raise ImportError("OK-VQA answer stemming needs `nltk` (with punkt, averaged_perceptron_tagger and wordnet data); "
                              "it is not installed in this environment") from e
        try:
            import inflection
        except ImportError as e:                                              # pragma: no cover
            raise ImportError("OK-VQA answer stemming needs the `inflection` package; it is not installed") from e
        if self._manual is None:
            path = os.environ.get("LICV_OKVQA_MANUAL_MATCHES")
            self._manual = json.load(open(path)) if path else {}
        self._nltk, self._inflection, self._lemmatizer = nltk, inflection, WordNetLemmatizer()

    def stem(self, text: str) -> str:
        """Word by word: exception table; "...ing" -> verb lemma; plural nouns (POS NNS / NNPS) -> singular."""
        self._load()
        out = []
        for word, pos in self._nltk.pos_tag(self._nltk.tokenize.word_tokenize(text)):
            if word in self._manual:
                word = self._manual[word]
            elif word.endswith("ing"):
                word = self._lemmatizer.lemmatize(word, "v")
            elif pos.startswith("NNS") or pos.startswith("NNPS"):
                word = self._inflection.singularize(word)
            out.append(word)
        return " ".join(out)


stemmer = OKVQAStemmer()


def postprocess_ok_vqa_generation(predictions: str) -> str:
    answer = re.split("Question|Answer|Short", predictions, 1)[0]
    answer = re.split(", ", answer, 1)[0]
    return stemmer.stem(answer)


def ok_vq_postprocess(text: str, model_name: str) -> Optional[str]:
    """ref:utils.py:128-133 (the reference's own spelling of the name)."""
    if "flamingo" in model_name:
        return postprocess_ok_vqa_generation(text).strip()
    if "idefics" in model_name:
        return postprocess_ok_vqa_generation(text).replace("\n", "").strip()
    return None
