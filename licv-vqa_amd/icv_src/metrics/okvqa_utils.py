"""OK-VQA answer post-processing (ref:icv_src/metrics/okvqa_utils.py:205-215, ref:utils.py:128-133): cut the generation at the
next prompt keyword / first ", ", then stem every word the way the OK-VQA annotations were stemmed.

The stemmer needs two packages this image does not have (no network): `nltk` with its tokenizer, POS-tagger and WordNet data,
and `inflection`.  They are imported when the first answer is stemmed; if either is missing the call raises an ImportError that
says which — it never falls back to an unstemmed answer, which would silently lower every OK-VQA score.

The per-word EXCEPTION TABLE of the OK-VQA v1.1 procedure (168 word -> stem pairs the reference checks before its "ing" and plural
rules, ref :15-184) ships as data next to this module, `okvqa_manual_matches.json` (written by tools/make_okvqa_table.py from the
reference checkout); LICV_OKVQA_MANUAL_MATCHES names another JSON file.  A missing table is an error, never an empty default.
The rule ORDER is pinned on the CPU with stand-in nltk / inflection modules (tests/test_okvqa_utils.py)."""
from __future__ import annotations

import json
import os
import re
from pathlib import Path
from typing import Dict, Optional

_TABLE = Path(__file__).with_name("okvqa_manual_matches.json")


def load_manual_matches() -> Dict[str, str]:
    path = Path(os.environ.get("LICV_OKVQA_MANUAL_MATCHES", _TABLE))
    if not path.exists():
        raise FileNotFoundError(f"OK-VQA stemming exception table {path} not found (tools/make_okvqa_table.py writes it): refusing to "
                                "stem without it, the scores would silently differ from the reference's")
    table = json.loads(path.read_text())
    if not isinstance(table, dict) or not table:
        raise ValueError(f"{path} does not hold a word -> stem table")
    return table


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
            self._manual = load_manual_matches()
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
