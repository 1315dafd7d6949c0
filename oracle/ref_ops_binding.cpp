// Binding shim (test infrastructure, NOT product code): exposes the reference's own CPU attention kernels, compiled from
// the sources where they lie under /root/reference/sgl-kernel/csrc/cpu/{decode,extend}.cpp, as torch.ops.sgl_ref.*
// so that tests can check the oracle restatement against them.  Schemas copied from the reference's registration
// (torch_extension_cpu.cpp:264-275); nothing here reimplements or stands in for reference code.
#include <ATen/ATen.h>
#include <torch/library.h>

void decode_attention_cpu(at::Tensor& query, at::Tensor& k_cache, at::Tensor& v_cache, at::Tensor& output, at::Tensor& key,
                          at::Tensor& value, at::Tensor& loc, at::Tensor& attn_logits, at::Tensor& req_to_token,
                          at::Tensor& req_pool_indices, at::Tensor& seq_lens, double sm_scale, double logit_cap);
void extend_attention_cpu(at::Tensor& q_extend, at::Tensor& k_extend, at::Tensor& v_extend, at::Tensor& o_extend,
                          at::Tensor& k_buffer, at::Tensor& v_buffer, at::Tensor& req_to_token, at::Tensor& req_pool_indices,
                          at::Tensor& seq_lens, at::Tensor& extend_seq_lens, at::Tensor& extend_start_loc,
                          int64_t max_len_extend, double sm_scale, double logit_cap);

TORCH_LIBRARY(sgl_ref, m) {
  m.def("decode_attention_cpu(Tensor query, Tensor k_cache, Tensor v_cahce, Tensor(a!) output, Tensor key, Tensor value, "
        "Tensor loc, Tensor attn_logits, Tensor req_to_token, Tensor req_pool_indices, Tensor seq_lens, float sm_scale, "
        "float logit_cap) -> ()");
  m.impl("decode_attention_cpu", c10::kCPU, &decode_attention_cpu);
  m.def("extend_attention_cpu(Tensor q_extend, Tensor k_extend, Tensor v_extend, Tensor(a!) o_extend, Tensor k_buffer, "
        "Tensor v_buffer, Tensor req_to_token, Tensor req_pool_indices, Tensor seq_lens, Tensor extend_seq_lens, "
        "Tensor extend_start_loc, int max_len_extend, float sm_scale, float logit_cap) -> ()");
  m.impl("extend_attention_cpu", c10::kCPU, &extend_attention_cpu);
}
