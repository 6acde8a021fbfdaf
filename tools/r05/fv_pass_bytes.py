# design bytes per cell of every kernel of a coupled pimpleFoamYade step (the arrays the kernel names, each once; FP64) and the
# SURVEY.md 8(d) row it belongs to.  8(d): F1 pre-coupling 96 + F2/F4 assembly 152 + 32 = 280 B/cell per step; per corrector F5/F6
# 16 + 128 + 104 + 80 + 80 + 64 + 32 = 504 B/cell; the Krylov iterations 128 B/cell each; particle phase 128 Np + 232 Nc.
reg("k_tile_caps", None, "particle")
reg("k_locate_deposit", None, "particle")
reg("k_locate<true, true>", None, "particle")
reg("k_tile_reduce<1>", None, "particle")
reg("k_tile_reduce<2>", None, "particle")
reg("k_force_gaussian", None, "particle")
reg("k_set_source_zero", 64, "particle")
reg("k_pre_coupling", 224, "F1 pre-coupling")            # U 24 p 8 alpha 8 -> gradP 24 divT 24 Uold 24 cellrec 64 (first call); second call: U 24 alpha 8 -> G 72
reg("k_pre_G_divG", 56, "F4 UcEqn")
reg("k_div_G", 96, "F4 UcEqn")
reg("k_interp_alpha_cells", 32, "F8 post-coupling prep")
reg("k_assemble_momentum<false>", 224, "F4 UcEqn")
reg("k_rAUf_phi_forces_cells", 80, "F4 UcEqn")
reg("k_bmom", 104, "F3 predictor")
reg("k_bmom_faces", 112, "F3 predictor")
reg("k_mom_pass", 128, "F3 predictor (Krylov)")
reg("k_sum3", 24, "F3 predictor (Krylov)")
reg("k_HbyA", 128, "F5 corrector")
reg("k_phiHbyA_cells<1>", 192, "F5 corrector")
reg("k_phiHbyA_cells<2>", 96, "F5 corrector")
reg("k_assemble_pressure<true>", 120, "F5 corrector")
reg("k_assemble_pressure<false>", 64, "F5 corrector")
reg("k_corr_front<true, true>", 168, "F5 corrector")      # HbyA 24 dcorr 24 rAU 8 alpha 8 phiForces 24 p 8 -> phiHbyA 24 A 32 rhs 8 r0 8
reg("k_corr_front<false, true>", 136, "F5 corrector")     # ... without the matrix store
reg("k_corr_back<true, true>", 144, "F5 corrector")       # p 8 phiHbyA 24 rAU 8 alpha 8 phiForces 24 HbyA 24 -> phi 24 U 24
reg("k_mom_pass<false>", 128, "F3 predictor (Krylov)")
reg("k_mom_pass<true>", 184, "F3 predictor (Krylov)")     # + src 24 rAU 8 -> HbyA 24: the first corrector's H-operator sweep rides on it
reg("k_flux_correct_cells", 152, "F5 corrector")
reg("k_U_correct<true>", 136, "F5 corrector")
reg("k_p_init", 56, "F6 pEqn solve")
for nm in ("k_mg_ref_term", "k_mg_coarsen", "k_mg_coarse_factor", "k_mg_smooth_two_from_zero", "k_mg_residual_restrict_tiled", "k_mg_residual_restrict",
           "k_mg_tail", "k_mg_smooth_prolong", "k_mg_smooth", "k_mg_smooth_dot", "k_p_apply_dot<false>", "k_p_apply_dot<true>",
           "k_pcg_cg_update<true>", "k_pcg_cg_update<false>",
           # (the two-cells-per-thread forms of the same sweeps)
           "k_mg_smooth_two_from_zero2", "k_mg_residual_restrict_tiled2", "k_mg_smooth_prolong2", "k_mg_smooth2", "k_mg_smooth_dot2", "k_p_apply_dot2<false>",
           "k_p_apply_dot2<true>", "k_pcg_cg_update2<true>", "k_pcg_cg_update2<false>", "k_p_apply2",
           "k_mg_coarse_factor<1>", "k_mg_coarse_factor<2>", "k_mg_coarse_factor<3>",
           "k_mg_residual_restrict_tiled2<64, 2, 2>", "k_mg_residual_restrict_tiled2<32, 4, 2>", "k_mg_residual_restrict_tiled2<16, 4, 4>", "k_mg_residual_restrict_tiled2<8, 8, 4>"):
    reg(nm, None, "F6 pEqn solve")
reg("k_reduce_finalize", None, "reductions")
reg("__amd_rocclr_fillBufferAligned", None, "fills")
