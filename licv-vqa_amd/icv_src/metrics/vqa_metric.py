"""VQA accuracy scoring of generated answers: closes the accuracy loop of ``inference.py`` for runs on real
checkpoints.  CPU string work; same entry points and results as ref:icv_src/metrics/vqa_metric.py:528-561 and ref:utils.py:122-126
(``compute_vqa_accuracy(result_json_path, question_json_path, annotation_json_path) -> {"overall", "perQuestionType",
"perAnswerType"}``, ``postprocess_vqa_generation``, ``vqa_postprocess``), which follow the official VQA evaluation:
answers are normalised (punctuation, digits/articles, contractions) and a prediction scores min(1, #matching humans / 3),
averaged over the ten leave-one-out subsets of the human answers.  Pinned by fixture g14 (the reference's own evaluator run on
a corpus of strings and on synthetic annotation files).  The OK-VQA stemmer (ref:icv_src/metrics/okvqa_utils.py, nltk) is not
included: nltk is not available offline.
"""
from __future__ import annotations

import json
import re
from typing import Dict, Iterable, List

# canonical contractions the official evaluation restores; the accepted misspellings are derived below: a word with one
# apostrophe is accepted without it, a word with several is accepted with any ONE of them missing
_CANONICAL = (
    "'ow's'at 'twas I'd've I'm I've ain't aren't can't could've couldn't couldn't've didn't doesn't don't hadn't hadn't've hasn't "
    "haven't he'd he'd've he's how'd how'll how's isn't it'd it'd've it'll ma'am might've mightn't mightn't've must've mustn't "
    "needn't not've o'clock oughtn't shan't she'd've should've shouldn't shouldn't've somebody'd've somebody'll somebody's "
    "someone'd someone'd've someone'll someone's something'd something'd've something'll that's there'd there'd've there're "
    "there's they'd they'd've they'll they're they've wasn't we'd've we've weren't what'll what're what's what've when's where'd "
    "where's where've who'd who'd've who'll who's who've why'll why're why's won't would've wouldn't wouldn't've y'all y'all'd've "
    "y'all'll you'd you'd've you'll you're you've").split()


def _contraction_table() -> Dict[str, str]:
    table = {}
    for word in _CANONICAL:
        cuts = [i for i, ch in enumerate(word) if ch == "'"]
        if len(cuts) == 1:
            table[word.replace("'", "")] = word
        else:
            for i in cuts:
                table[word[:i] + word[i + 1:]] = word
    # three entries of the official table that follow no rule (two identities, one written the other way round)
    table.update({"let's": "let's", "she's": "she's", "somebody'd": "somebodyd"})
    return table


_NUMBER_WORDS = dict(zip("none zero one two three four five six seven eight nine ten".split(), "0 0 1 2 3 4 5 6 7 8 9 10".split()))
_ARTICLES = ("a", "an", "the")
_PUNCT = [";", "/", "[", "]", '"', "{", "}", "(", ")", "=", "+", "\\", "_", "-", ">", "<", "@", "`", ",", "?", "!"]
_PERIOD = re.compile(r"(?!<=\d)(\.)(?!\d)")
_DIGIT_COMMA = re.compile(r"(\d)(\,)(\d)")


class AnswerNormalizer:
    contractions = _contraction_table()

    def punctuation(self, text: str) -> str:
        out = text
        has_digit_comma = _DIGIT_COMMA.search(text) is not None
        for p in _PUNCT:
            # decided on the ORIGINAL text: a mark next to a space (or any mark when a digit,digit comma is present) is deleted,
            # otherwise it becomes a space
            out = out.replace(p, "" if (p + " " in text or " " + p in text or has_digit_comma) else " ")
        # the official code passes re.UNICODE in the `count` position of sub(): at most 32 periods are stripped; kept as is
        return _PERIOD.sub("", out, int(re.UNICODE))

    def digits_articles(self, text: str) -> str:
        words = [_NUMBER_WORDS.get(w, w) for w in text.lower().split()]
        words = [w for w in words if w not in _ARTICLES]
        return " ".join(self.contractions.get(w, w) for w in words)

    def __call__(self, text: str) -> str:
        text = text.replace("\n", " ").replace("\t", " ").strip()
        return self.digits_articles(self.punctuation(text))


def question_accuracy(prediction: str, human_answers: Iterable[str], norm: AnswerNormalizer = AnswerNormalizer()) -> float:
    """Accuracy of one prediction against the human answers of its question (each answer in turn is left out)."""
    pred = norm(prediction)
    humans = [norm(a) for a in human_answers]
    scores = []
    for i in range(len(humans)):
        others = humans[:i] + humans[i + 1:]
        scores.append(min(1.0, sum(a == pred for a in others) / 3.0))
    return sum(scores) / len(scores)


def compute_vqa_accuracy(result_json_path, question_json_path, annotation_json_path, n: int = 2) -> Dict:
    """Scores every question of the RESULT file; returns the reference's accuracy dict (percentages rounded to n places)."""
    annotations = json.load(open(annotation_json_path))["annotations"]
    json.load(open(question_json_path))                                   # must exist and parse, as for the reference's loader
    results = json.load(open(result_json_path)) if isinstance(result_json_path, str) else result_json_path
    assert isinstance(results, list), "results is not an array of objects"
    by_q = {a["question_id"]: a for a in annotations}
    norm = AnswerNormalizer()
    overall: List[float] = []
    per_qt: Dict[str, List[float]] = {}
    per_at: Dict[str, List[float]] = {}
    last = {r["question_id"]: r for r in results}                          # the reference indexes results by question id: a repeated id keeps its last result
    for qid in dict.fromkeys(r["question_id"] for r in results):
        ann = by_q[qid]
        # identical human-answer records (same answer, id and confidence) count as the same person in the leave-one-out
        records = ann["answers"]
        humans = [norm(a["answer"]) for a in records]
        pred = norm(last[qid]["answer"])
        scores = []
        for i, rec in enumerate(records):
            others = [humans[j] for j, o in enumerate(records) if _record(o, humans[j]) != _record(rec, humans[i])]
            scores.append(min(1.0, sum(a == pred for a in others) / 3.0))
        acc = sum(scores) / len(scores)
        overall.append(acc)
        per_qt.setdefault(ann["question_type"], []).append(acc)
        per_at.setdefault(ann.get("answer_type", "other"), []).append(acc)
    pct = lambda xs: round(100 * float(sum(xs)) / len(xs), n)
    return {"overall": pct(overall), "perQuestionType": {k: pct(v) for k, v in per_qt.items()},
            "perAnswerType": {k: pct(v) for k, v in per_at.items()}}


def _record(rec: dict, normalised: str):
    return tuple(sorted((k, normalised if k == "answer" else v) for k, v in rec.items()))


def postprocess_vqa_generation(predictions: str) -> str:
    """Cut a generation at the next prompt keyword and at the first ", " (ref:icv_src/metrics/vqa_metric.py:558-561)."""
    answer = re.split("Question|Answer|Short", predictions, 1)[0]
    return re.split(", ", answer, 1)[0]


def vqa_postprocess(text: str, model_name: str) -> str:
    """ref:utils.py:122-126."""
    if "flamingo" in model_name:
        return postprocess_vqa_generation(text).strip()
    if "idefics" in model_name:
        return postprocess_vqa_generation(text).replace("\n", "").strip()
    return None
