"""API surface of the reference's ``models/videollama.py:1-24``.

VideoLLaMA2 is an opaque third-party model (``videollama2`` package, git branch ``audio_visual``)
with no Q-Former seam on the hot path this build covers.  The class keeps
the reference's contract so callers written against it keep working: ``VideoLLaMA(path)``,
``.processor``, ``.generate(samples)`` returning ONE string built from batch element 0 and never
raising (any failure -> ``"error"``).
"""


class VideoLLaMA:
    def __init__(self, path):
        from videollama2 import model_init  # third-party; absent offline -> ImportError, as in the reference
        self.model, self.processor, self.tokenizer = model_init(path)

    def generate(self, samples):
        try:
            from videollama2 import mm_infer
            output = mm_infer(samples["video"][0], samples["text_input"][0], model=self.model,
                              tokenizer=self.tokenizer, modal="video", do_sample=False)
        except Exception:  # the reference swallows every error here (models/videollama.py:21-23)
            print("generation error")
            output = "error"
        return output
