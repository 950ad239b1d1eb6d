/* rt_shade_phase.inl — the second half of a wave iteration of trace_body (rt_kernels.h): the rest of one iteration of Trace's bounce loop
 * (RC:488-538) for the lanes whose intersection is complete.  Textually included at the two places it can follow the first half: directly
 * (every variant but the pooled FLAT one — the code is then exactly the nested block of rounds 1-5, a callable changed the register
 * allocation of the > 64-model variant), or behind the chain exchange that the pooled FLAT variant runs between the halves with every
 * lane of the wave present.  Not a header: it reads and writes trace_body's locals. */
        if (inTrav && (FLAT || traverse<STATS, true, MANY, HOT>(a, rpos, rdir, stackBase, extBase, h, t, st, hotLds, hotUnits))) {
            inTrav = false;
            /* the rest of one iteration of Trace's bounce loop — RC:488-538 */
            bool endPath = false;
            if (h.obj < 0) {
                phase_mark<STATS>(st, PH_SKY);
                const RT_CAS KArgs& c = cold_args();
                if (c.useSky) pathLight = pathLight + transmittance * environment_light(c, rdir);
                endPath = true;
            } else {
                /* resolve the winner: position, normal, material */
                phase_mark<STATS>(st, PH_SHADE_HIT);
                rt_f3 hpos, normal;
                resolve_hit(a, rpos, rdir, h, hpos, normal);
                const DMaterial mat = a.materials[h.obj];

                /* The glass (RC:499-518) and opaque (RC:519-533) branches both draw
                 * diffuseDir = normalize(normal + RandomDirection) — the costliest piece
                 * (3 log, 3 cos, 4 sqrt).  It is hoisted so that all hit lanes execute it
                 * together; each lane still consumes its random numbers in its branch's
                 * order: opaque = [isSpecular, direction x6], glass = [direction x6, choice]. */
                const bool isGlass = mat.flag == RT_MATERIAL_GLASS;
                float uSpec = 0.0f;
                if (!isGlass) uSpec = rt_random_value(&rng); /* RC:521 */
                const rt_f3 diffuseDir = rt_normalize(normal + rand_direction(&rng)); /* RC:509 / RC:525 */
                /* Both branches end in normalize(lerp(A, B, t)) of a direction pair: opaque (diffuseDir, reflect, smoothness x
                 * isSpecular), glass either (diffuseDir, reflect, specularProbability) or (-diffuseDir, refract, smoothness).  The
                 * reference normalises both glass candidates and keeps one (RC:511-516); only the kept one is observable, so the
                 * branches just pick (A, B, t) and ONE lerp + normalize follows for all hit lanes. */
                const rt_f3 specularDir = rt_reflect(rdir, normal); /* == the glass branch's reflectDir, RC:419-422 */
                rt_f3 lerpA = diffuseDir, lerpB = specularDir;
                float lerpT;
                if (isGlass) {
                    phase_mark<STATS>(st, PH_GLASS);
                    if (h.backface) { /* RC:502 */
                        rt_f3 e = ((-h.dst) * rt_v3(mat.absorption[0], mat.absorption[1], mat.absorption[2])) * mat.absorptionStrength;
                        transmittance = transmittance * rt_v3(rt_exp(e.x), rt_exp(e.y), rt_exp(e.z));
                    }
                    float iorCurrent = h.backface ? mat.ior : 1.0f;
                    float iorNext = h.backface ? 1.0f : mat.ior;
                    const rt_f3 refractDir = refract_dir(rdir, normal, iorCurrent, iorNext);
                    const float reflectWeight = reflectance(rdir, normal, iorCurrent, iorNext);
                    const bool followReflection = rt_random_value(&rng) <= reflectWeight; /* RC:515 */
                    lerpT = mat.specularProbability;
                    if (!followReflection) {
                        lerpA = -diffuseDir;
                        lerpB = refractDir;
                        lerpT = mat.smoothness;
                    }
                } else {
                    const bool isSpecular = mat.specularProbability >= uSpec;
                    lerpT = mat.smoothness * (isSpecular ? 1.0f : 0.0f);
                    rt_f3 emitted = rt_v3(mat.emissionCol[0], mat.emissionCol[1], mat.emissionCol[2]) * mat.emissionStrength;
                    pathLight = pathLight + emitted * transmittance;
                    transmittance = transmittance * material_colour(mat, hpos, normal, isSpecular);
                }
                rdir = rt_normalize(rt_lerp3(lerpA, lerpB, lerpT));
                rpos = isGlass ? hpos + (0.001f * normal) * rt_sign(rt_dot(normal, rdir)) : hpos + (normal * 0.001f);
                /* RC:535-538 Russian roulette */
                float p = rt_max(transmittance.x, rt_max(transmittance.y, transmittance.z));
                if (rt_random_value(&rng) >= p) {
                    endPath = true;
                } else {
                    transmittance = transmittance * rt_rcp(p);
                    if (MANY) {
                        const uint32_t b = extBase[(1 + a.extWords) * RT_WAVE] + 1u;
                        extBase[(1 + a.extWords) * RT_WAVE] = b;
                        if ((int)b > a.maxBounce) endPath = true;
                    } else {
                        bounce++;
                        if (bounce > a.maxBounce) endPath = true; /* RC:485: i <= MaxBounceCount */
                    }
                }
            }
            if (endPath) {
                PXF(PX_TIX) = PXF(PX_TIX) + pathLight.x; /* RC:578: totalIncomingLight += Trace(...) */
                PXF(PX_TIY) = PXF(PX_TIY) + pathLight.y;
                PXF(PX_TIZ) = PXF(PX_TIZ) + pathLight.z;
                pathActive = false;
            }
        
        }
