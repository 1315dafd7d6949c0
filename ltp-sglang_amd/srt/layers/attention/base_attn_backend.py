"""AttentionBackend: the abstract plugin surface every attention backend implements
(python/sglang/srt/layers/attention/base_attn_backend.py:14-117).  The reference selects a backend with a
string switch (model_runner.py:1374-1460); anything that subclasses this ABC, is built from a ModelRunner-shaped
object and is assigned to ``forward_batch.attn_backend`` drops in."""
from abc import ABC, abstractmethod


class AttentionBackend(ABC):
    @abstractmethod
    def init_forward_metadata(self, forward_batch):
        """Called once per batch before the layers run (model_runner.py:1542,1562)."""
        raise NotImplementedError()

    # ---- graph-capture hooks (cuda_graph_runner.py:280,618,760) ----
    def init_cuda_graph_state(self, max_bs: int, max_num_tokens: int):
        raise NotImplementedError()

    def init_forward_metadata_capture_cuda_graph(self, bs, num_tokens, req_pool_indices, seq_lens, encoder_lens,
                                                 forward_mode, spec_info):
        raise NotImplementedError()

    def init_forward_metadata_replay_cuda_graph(self, bs, req_pool_indices, seq_lens, seq_lens_sum, encoder_lens,
                                                forward_mode, spec_info, seq_lens_cpu):
        raise NotImplementedError()

    def get_cuda_graph_seq_len_fill_value(self):
        raise NotImplementedError()

    def forward(self, q, k, v, layer, forward_batch, save_kv_cache=True, **kwargs):
        """Mode dispatch: IDLE -> empty [T, Hq*Dv]; DECODE -> forward_decode; otherwise forward_extend."""
        mode = forward_batch.forward_mode
        if mode.is_idle():
            return q.new_empty(q.shape[0], layer.tp_q_head_num * layer.v_head_dim)
        fn = self.forward_decode if mode.is_decode() else self.forward_extend
        return fn(q, k, v, layer, forward_batch, save_kv_cache=save_kv_cache, **kwargs)

    def forward_decode(self, q, k, v, layer, forward_batch, save_kv_cache=True):
        raise NotImplementedError()

    def forward_extend(self, q, k, v, layer, forward_batch, save_kv_cache=True):
        raise NotImplementedError()

    def support_triton(self):
        """False switches the reference's host helpers to their torch variants
        (forward_batch_info.py:423-432, schedule_batch.py:1292-1310)."""
        return True
