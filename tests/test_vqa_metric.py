"""VQA accuracy scoring against what the reference's own evaluator produced (fixture g14)."""
import json

from icv_src.metrics.vqa_metric import (AnswerNormalizer, compute_vqa_accuracy, postprocess_vqa_generation, question_accuracy,
                                        vqa_postprocess)


def test_normalisation_matches_reference_on_the_corpus(golden):
    z = golden("g14_vqa_metric")
    norm = AnswerNormalizer()
    for text, punct, full in zip(z["corpus"], z["punct"], z["full"]):
        t = str(text).replace("\n", " ").replace("\t", " ").strip()
        assert norm.punctuation(t) == str(punct), repr(text)
        assert norm(str(text)) == str(full), repr(text)


def test_generation_postprocess_matches_reference(golden):
    z = golden("g14_vqa_metric")
    for a, b in zip(z["gen_in"], z["gen_out"]):
        assert postprocess_vqa_generation(str(a)) == str(b)
    assert vqa_postprocess(" red\nQuestion: x", "idefics-9b") == "red" and vqa_postprocess("x", "other-model") is None


def test_accuracy_on_synthetic_files_matches_reference(golden, tmp_path):
    z = golden("g14_vqa_metric")
    paths = {}
    for k in ("ann", "que", "res"):
        paths[k] = str(tmp_path / f"{k}.json")
        open(paths[k], "w").write(str(z[f"file_{k}"]))
    assert compute_vqa_accuracy(paths["res"], paths["que"], paths["ann"]) == json.loads(str(z["accuracy"]))
    assert question_accuracy("Two.", ["2", "two", "2", "three", "2", "a 2", "one", "2", "2", "2"]) == 1.0
    assert abs(question_accuracy("three", ["2", "two", "2", "three", "2", "a 2", "one", "2", "2", "2"]) - 0.3) < 1e-12
